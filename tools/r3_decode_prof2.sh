#!/bin/bash
# per-kernel times of the full-size decode (default settings), every kernel listed; the last decode's timeline with its gaps
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_dec_r3b; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o dec --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_fullsize.py > $out/log.txt 2>&1
echo "rc=$?"; tail -1 $out/log.txt
python3 - $out <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/decode_kernels_r3b.txt
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
acc=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].split("(")[0][:70]
    acc[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
    print("  %-72s n=%3d avg %8.1f us  total %9.1f" % (n, len(v), sum(v)/len(v), sum(v)))
# timeline of the last decode: from the last k_find / first kernel after the biggest gap
last=[i for i,r in enumerate(rows) if "k_par_count" in r["Kernel_Name"]][-1]
i0=last
while i0>0 and int(rows[i0]["Start_Timestamp"])-int(rows[i0-1]["End_Timestamp"])<2_000_000: i0-=1
t0=int(rows[i0]["Start_Timestamp"]); prev=t0
print("timeline of the last decode (us from its first kernel: start, duration, gap before)")
for r in rows[i0:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("  %9.1f %8.1f %7.1f  %s" % ((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,r["Kernel_Name"].split("(")[0][:60]))
    prev=e
PY
