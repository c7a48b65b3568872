#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4p8; mkdir -p $O
MIJ_PX_DEBUG=1 timeout -k 10 900 python3 tools/r4_px_test.py > $O/px_test.txt 2>&1; echo "rc=$?"; grep -v "^\[px\]" $O/px_test.txt | tail -30; grep -c "FELL BACK" $O/px_test.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o px --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/r4_px_prof.py 2048 95 1 3 > $O/prof_log.txt 2>&1; echo "prof rc=$?"; grep "rep\|file" $O/prof_log.txt
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4p8"
f = glob.glob(O + "/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    out = ["%-70s calls %5s total %9.3f ms avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3) for r in rows[:25]]
    open(O + "/kernel_stats_top.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out))
PY
rm -rf $O/prof
