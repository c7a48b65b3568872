#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/ into profiles/ under this round's names.
    python tools/collect_profiles.py r03"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [("refresh/lines/n1.json", "bench_n1.json"), ("refresh/lines/fixed.json", "bench_n1_fixed_huffman.json"), ("refresh/lines/two.json", "bench_n1_two_streams.json"),
         ("refresh/bench_prog.json", "bench_n1_progressive.json"), ("refresh/bench_440_cpu.json", "bench_n1_440_ijg_cpu_leg.json"),
         ("refresh/table1.jsonl", "table1_samplings.jsonl"), ("refresh/hbm_traffic.json", "hbm_traffic.json"), ("refresh/pmc_sq_summary.txt", "pmc_sq_summary.txt"),
         ("refresh/decode_fullsize.json", "decode_fullsize.json"), ("refresh/secondary_fullsize.json", "secondary_fullsize.json")]
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s):
        shutil.copy(s, os.path.join(P, "%s_%s" % (tag, dst)))
        print("copied", src)
    else:
        print("MISSING", src)
for pat, dst in (("prof_bench/**/*kernel_stats.csv", "bench_kernel_stats.csv"), ("prof_decode/**/*kernel_stats.csv", "decode_kernel_stats.csv"),
                 ("prof_prog1/**/*kernel_stats.csv", "progressive_one_stream_kernel_stats.csv"), ("pmc_fetch/**/*counter_collection.csv", "pmc_fetch_size.csv"),
                 ("pmc_write/**/*counter_collection.csv", "pmc_write_size.csv")):
    fs = glob.glob(os.path.join(G, pat), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(P, "%s_%s" % (tag, dst)))
        print("copied", os.path.relpath(fs[0], G))
    else:
        print("MISSING", pat)
