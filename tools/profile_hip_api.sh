#!/bin/bash
# rocprofv3 HIP-API trace + stats of a short default bench run: which runtime calls a steady-state step makes (no --pmc here)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_hip
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --hip-trace --stats -d $out -o api --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-psnr > $out/log.txt 2>&1
echo "profile rc=$?"
