#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>   -- one rocprofv3 --pmc pass over a short bench run; output under gpurun_out/pmc_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out     # stale counter files from an earlier pass would be averaged in
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $out -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --kernel-pass 2 --stage-a-pass 2 --settle-rounds 0 --no-cpu-baseline --no-psnr --no-extra > $out/log.txt 2>&1
echo "pass $tag rc=$?"
