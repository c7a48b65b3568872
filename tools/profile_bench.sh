#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (default --steps / --warmup; only the CPU-baseline and PSNR legs, which
# launch no kernels, are switched off); output under gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-psnr > $out/log.txt 2>&1
echo "profile rc=$?"; tail -1 $out/log.txt | cut -c1-300
