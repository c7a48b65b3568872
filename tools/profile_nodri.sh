#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_nodri
rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o nodri --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_nodri_bench.py 1000 > $out/log.txt 2>&1
echo "profile rc=$?"; tail -1 $out/log.txt | cut -c1-400
