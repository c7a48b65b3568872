#!/bin/bash
# rocprofv3 kernel trace + stats of a full-size decode; output under gpurun_out/prof_decode/
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_decode
mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out -o dec --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_fullsize.py > $out/log.txt 2>&1
echo "profile rc=$?"; tail -2 $out/log.txt
