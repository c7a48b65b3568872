#!/bin/bash
# per-kernel times of the full-size decode for a few settings of the speculative tail / the side stream
cd /tmp && export TMPDIR=/tmp
# (the side-stream form measured with this script in round 3 is no longer in the library; what remains are the speculative tail
#  MIJ_PAR_TAIL and the sparse-pass threshold MIJ_PAR_SPARSE)
for cfg in "1024 0" "1024 256" "0 256"; do
  set -- $cfg
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_dec_$1_$2; rm -rf $out; mkdir -p $out
  export MIJ_PAR_SPARSE=$1 MIJ_PAR_TAIL=$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o dec --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_fullsize.py > $out/log.txt 2>&1
  echo "== sparse_max=$1 tail=$2 rc=$?"; python3 - $out <<'PY'
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
