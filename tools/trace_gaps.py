#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 --kernel-trace CSV (tools/trace_gaps.sh): prints the steady-state sequence with the gap
in front of every kernel and the busy / idle totals per image."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path))]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")) for r in rows), key=lambda x: x[0])
# steady state: the last 6 k_transform launches
idx = [i for i, k in enumerate(ks) if "k_transform" in k[2]]
lo, hi = idx[-7], idx[-1]
seq = ks[lo:hi]
busy_end = seq[0][0]
idle = 0
for s, e, n, q in seq:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print("%8.1f us gap  %8.1f us  q%-3s %s" % (max(gap, 0) / 1e3 if gap > 0 else gap / 1e3, (e - s) / 1e3, q, n))
    busy_end = max(busy_end, e)
span = seq[-1][1] - seq[0][0]
print("span %.1f us for 6 images = %.1f us per image; idle %.1f us per image" % (span / 1e3, span / 6e3, idle / 6e3))
