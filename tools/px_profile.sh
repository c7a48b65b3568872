#!/bin/bash
# A full-size (or any-height) progressive no-DRI decode under rocprofv3: per-kernel totals and the busiest queue's sequence.
# Usage: bash tools/px_profile.sh HEIGHT REPS nocheck [Q SS]     (arguments of tools/decode_prog_nodri_fullsize.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/pxprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 rocprofv3 --kernel-trace --stats -d $O/prof -o px --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_prog_nodri_fullsize.py "$@" > $O/prof_log.txt 2>&1; echo "prof rc=$?"; grep "progress\|^{" $O/prof_log.txt | cut -c1-300
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pxprof"
f = glob.glob(O + "/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:24]:
        print("%-64s calls %6s total %10.3f ms avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
t = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)
if t:
    rows = list(csv.DictReader(open(t[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
    byq = {}
    for r in rows:
        q = r["Queue_Id"]; byq.setdefault(q, [0, 0])
        byq[q][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); byq[q][1] += 1
    print("whole run: %.1f ms from first to last kernel" % ((t1 - t0) / 1e6))
    for q, (ns, n) in byq.items():
        print(" queue %s: %.2f ms busy in %d kernels" % (q, ns / 1e6, n))
PY
rm -rf $O/prof
