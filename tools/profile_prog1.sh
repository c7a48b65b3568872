#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_prog1
rm -rf $out; mkdir -p $out
export MIJ_PROG_STREAMS=1 MIJ_PROG_ORDER=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o prog --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --progressive --steps 3 --warmup 1 --no-cpu-baseline --no-psnr > $out/log.txt 2>&1
echo "profile rc=$?"
