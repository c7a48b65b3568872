#!/bin/bash
# kernel trace of a short bench run, for tools/trace_gaps.py (idle time between kernels); extra bench flags in "$@"
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$1; shift
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-psnr "$@" > $out/log.txt 2>&1
echo "trace rc=$?"
