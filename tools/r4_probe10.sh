#!/bin/bash
# full-size progressive no-DRI decode under rocprofv3: per-kernel totals
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4p10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/prof -o px --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_prog_nodri_fullsize.py 40000 2 nocheck > $O/prof_log.txt 2>&1; echo "prof rc=$?"; grep "progress\|^{" $O/prof_log.txt | cut -c1-300
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4p10"
f = glob.glob(O + "/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    out = ["%-64s calls %5s total %9.3f ms avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3) for r in rows[:22]]
    open(O + "/kernel_stats_top.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out))
# timeline of the second decode: per stream busy time
t = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)
if t:
    rows = list(csv.DictReader(open(t[0])))
    rows = [r for r in rows if "k_px" in r["Kernel_Name"] or "k_unstuff" in r["Kernel_Name"] or "k_idct" in r["Kernel_Name"] or "k_dc_refine" in r["Kernel_Name"] or "exscan" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    half = rows[len(rows) // 2:]
    t0 = int(half[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in half)
    print("second decode: %.2f ms wall over its kernels" % ((t1 - t0) / 1e6))
    byq = {}
    for r in half:
        q = r["Queue_Id"]; byq.setdefault(q, [0, 0])
        byq[q][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); byq[q][1] += 1
    for q, (ns, n) in byq.items():
        print(" queue %s: %.2f ms busy in %d kernels" % (q, ns / 1e6, n))
    qmax = max(byq, key=lambda q: byq[q][0])
    seq = [r for r in half if r["Queue_Id"] == qmax]
    lines = []
    for r in seq:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d >= float(os.environ.get("TL_MIN_MS", "0.25")):
            lines.append("  %8.2f ms at %8.2f ms %s grid %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e6, r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    open(O + "/busiest_queue_sequence.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:500]))
PY
rm -rf $O/prof
