#!/bin/bash
# per-kernel times of the full-size decode for a few settings of the speculative tail / the side stream
cd /tmp && export TMPDIR=/tmp
for cfg in "1 0" "1 256" "0 256"; do
  set -- $cfg
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_dec_$1_$2; rm -rf $out; mkdir -p $out
  if [ "$1" = "1" ]; then export MIJ_PAR_SERIAL=1; else unset MIJ_PAR_SERIAL; fi
  export MIJ_PAR_TAIL=$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o dec --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_fullsize.py > $out/log.txt 2>&1
  echo "== serial=$1 tail=$2 rc=$?"; python3 - $out <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
acc=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].split("(")[0][:60]
    acc[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
    if "par_decode" in n or "idct" in n or "upsample" in n or "clean" in n or "exscan" in n:
        print("  %-62s n=%3d avg %8.1f us  total %9.1f" % (n, len(v), sum(v)/len(v), sum(v)))
PY
done
