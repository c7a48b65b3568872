#!/bin/bash
# usage: tools/pmc_decode.sh <tag> <counters...>   -- one rocprofv3 --pmc pass over the full-size decode (tools/decode_fullsize.py)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_dec_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $out -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_fullsize.py > $out/log.txt 2>&1
echo "pass $tag rc=$?"
