#!/usr/bin/env python3
"""Copies what tools/r4_refresh.sh left under gpurun_out/ into profiles/ under round 4's names."""
import csv, glob, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [("r4refresh/bench_driver_cmd.json", "bench_n1_driver_command.json"), ("r4refresh/bench_200.json", "bench_n1.json"),
         ("r4refresh/bench_one_stream.json", "bench_n1_one_stream_loop.json"), ("r4refresh/bench_tables_ahead.json", "bench_n1_tables_ahead_loop.json"),
         ("r4refresh/bench_fixed.json", "bench_n1_fixed_huffman.json"), ("r4refresh/bench_prog.json", "bench_n1_progressive.json"),
         ("r4refresh/table1.jsonl", "table1_samplings.jsonl"), ("r4refresh/hbm_traffic.json", "hbm_traffic.json"), ("r4refresh/pmc_sq_summary.txt", "pmc_sq_summary.txt"),
         ("r4refresh/decode_fullsize.json", "decode_fullsize.json"), ("r4refresh/decode_prog_nodri_fullsize.json", "decode_progressive_nodri_fullsize.json"),
         ("r4refresh/secondary_fullsize.json", "secondary_fullsize.json"), ("r4refresh/px_cases.txt", "decode_progressive_nodri_cases.txt")]
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s):
        shutil.copy(s, os.path.join(P, "r04_" + dst)); print("copied", src)
    else:
        print("MISSING", src)
tl = os.path.join(G, "r4refresh/px_timeline.txt")
if os.path.exists(tl):
    body = open(tl).read()
    open(os.path.join(P, "r04_decode_progressive_kernel_stats.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats over tools/decode_prog_nodri_fullsize.py 40000 2 nocheck (tools/r4_probe10.sh, final library of round 4):\n"
        "per-kernel totals over both decodes, then the second decode's kernels (>= 0.25 ms) on its busiest queue = the luma chain\n" + body)
    print("copied px_timeline")
for pat, dst in (("prof_bench/**/*kernel_stats.csv", "bench_kernel_stats.csv"), ("prof_bench_one/**/*kernel_stats.csv", "bench_kernel_stats_one_stream_loop.csv"),
                 ("pmc_fetch/**/*counter_collection.csv", "pmc_fetch_size.csv"), ("pmc_write/**/*counter_collection.csv", "pmc_write_size.csv")):
    fs = glob.glob(os.path.join(G, pat), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(P, "r04_" + dst)); print("copied", os.path.relpath(fs[0], G))
    else:
        print("MISSING", pat)
# the one-stream pass inside the DEFAULT command: the launches of k_transform / k_encode / k_compact that ran alone (the kernel pass follows the timed
# loop: the last 12 of each in the trace, 2 of them warm-up) -- what roofline.avg_launch_ms of the default line must agree with
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
    open(os.path.join(P, "r04_bench_kernel_stats_per_kernel_pass.csv"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
