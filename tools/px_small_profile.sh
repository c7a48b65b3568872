#!/bin/bash
# kernel trace of a small progressive decode: launches, busy time and span of the last decode
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pxsmall; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof -o px --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/px_small_profile.py "$@" > $O/log.txt 2>&1; echo "rc=$?"; grep "^decode" $O/log.txt
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pxsmall"
t = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(t[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode = everything behind the last k_idct-type kernel but one
ends = [i for i, r in enumerate(rows) if "k_idct" in r["Kernel_Name"] or "k_upsample" in r["Kernel_Name"]]
cut = ends[-3] + 1 if len(ends) >= 3 else 0          # (idct + upsample per decode, or one fused kernel)
last = rows[cut:]
# keep only what follows the previous decode's final kernel
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print("last decode: %d kernel launches, span %.2f ms, sum of kernel durations %.2f ms" % (len(last), (t1 - t0) / 1e6, busy / 1e6))
c = collections.Counter(); d = collections.Counter()
for r in last:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    c[n] += 1; d[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, k in d.most_common(14):
    print("  %-42s %4d launches %8.3f ms" % (n, c[n], k / 1e6))
# per queue busy
q = collections.Counter()
for r in last: q[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("  busy per queue (ms):", {k: round(v / 1e6, 2) for k, v in q.items()})
PY
