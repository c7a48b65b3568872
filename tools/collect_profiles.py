#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/ into profiles/ under the round's names.  Usage: collect_profiles.py r05"""
import csv, glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
pairs = [("refresh/bench_driver_cmd.json", "bench_n1_driver_command.json"), ("refresh/bench_200.json", "bench_n1.json"),
         ("refresh/bench_fixed.json", "bench_n1_fixed_huffman.json"), ("refresh/bench_prog.json", "bench_n1_progressive.json"),
         ("refresh/bench_prog_3ranks_one_device.json", "bench_progressive_3ranks_one_device_rehearsal.json"),
         ("refresh/hbm_traffic.json", "hbm_traffic.json"), ("refresh/pmc_sq_summary.txt", "pmc_sq_summary.txt"),
         ("refresh/px_cases.txt", "decode_progressive_nodri_cases.txt"), ("refresh/px_thin.txt", "decode_progressive_thin_fullsize.jsonl")]
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s):
        shutil.copy(s, os.path.join(P, tag + "_" + dst)); print("copied", src)
    else:
        print("missing", src)
for pat, dst in (("prof_bench/**/*kernel_stats.csv", "bench_kernel_stats.csv"), ("pmc_fetch/**/*counter_collection.csv", "pmc_fetch_size.csv"),
                 ("pmc_write/**/*counter_collection.csv", "pmc_write_size.csv")):
    fs = glob.glob(os.path.join(G, pat), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(P, tag + "_" + dst)); print("copied", os.path.relpath(fs[0], G))
    else:
        print("missing", pat)
# the one-stream pass inside the DEFAULT command: the launches of k_transform / k_encode / k_compact that ran alone (the kernel pass follows the timed
# loop) -- what roofline.avg_launch_ms of the default line must agree with
t = glob.glob(os.path.join(G, "prof_bench/**/*kernel_trace.csv"), recursive=True)
if t:
    rows = list(csv.DictReader(open(t[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = ["kernel,launches,avg_us,min_us,max_us,which"]
    for key, name in (("k_transform<2, 1, true, true>", "k_transform (with statistics)"), ("k_encode<", "k_encode"), ("k_compact", "k_compact"), ("k_transform<2, 1, true, false>", "k_transform without statistics (stage A alone)")):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
        if key.endswith("false>"):
            sel, which = d[4:], "stage-A pass: all launches but the 4 warm-up ones"
        elif "k_transform" in key:
            sel, which = d[-15:-5], "per-kernel pass: its 10 timed launches (one image at a time on one stream, after the timed loop; the 5 launches behind them are the single-image timings)"
        else:      # behind the per-kernel pass these kernels also run 14 times in the stage-A pass (fixed-table encoder) and 5 times in the single-image timings
            sel, which = d[-29:-19], "per-kernel pass: its 10 timed launches (one image at a time on one stream, after the timed loop)"
        if sel:
            out.append('"%s",%d,%.1f,%.1f,%.1f,"%s"' % (name, len(sel), sum(sel) / len(sel), min(sel), max(sel), which))
        if d:
            out.append('"%s",%d,%.1f,%.1f,%.1f,"%s"' % (name, len(d), sum(d) / len(d), min(d), max(d), "ALL launches of the command (in the overlapped loop a kernel shares the device with another image's kernels)"))
    open(os.path.join(P, tag + "_bench_kernel_stats_per_kernel_pass.csv"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
