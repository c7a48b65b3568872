#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4p4; mkdir -p $O
python3 tools/r4_hiccup.py 400 > $O/hiccup.txt 2>&1; echo rc=$?; tail -30 $O/hiccup.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace -d $O/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/r4_hiccup.py 300 > $O/hiccup_traced.txt 2>&1; echo rc=$?
ls -la $O/trace/* | head; 
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r4p4"
kt = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# busy intervals -> gaps > 1 ms
end = int(rows[0]["End_Timestamp"])
out = []
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - end > 1_000_000:
        out.append("gap of %.2f ms at %.2f ms before %s" % ((s - end) / 1e6, (s - t0) / 1e6, r["Kernel_Name"][:60]))
    end = max(end, e)
open(O + "/gaps.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
# long kernels
long = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - t0) / 1e6, r["Kernel_Name"][:50]) for r in rows]
long.sort(reverse=True)
open(O + "/long_kernels.txt", "w").write("\n".join("%.3f ms at %.2f ms %s" % (d / 1e6, t, n) for d, t, n in long[:60]) + "\n")
ht = glob.glob(O + "/trace/**/*hip_api_trace.csv", recursive=True)
if ht:
    rows = list(csv.DictReader(open(ht[0])))
    slow = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"]) for r in rows]
    slow.sort(reverse=True)
    open(O + "/slow_api.txt", "w").write("\n".join("%.3f ms at %.2f ms %s" % (d / 1e6, t, n) for d, t, n in slow[:80]) + "\n")
PY
rm -rf $O/trace
