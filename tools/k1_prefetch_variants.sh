#!/bin/bash
# Round 4: K1 variants on one box: prefetch of the next pass's rows on / off, 4:4:4 chroma in registers / through LDS.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k1; mkdir -p $O; cd $R
V=${1:-default,no_prefetch}; shift
timeout -k 10 300 python3 -m pytest tests/test_gpu_encode.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt || exit 1
for css in "$@"; do
  for opt in "" "--no-optimize"; do
    echo "== css $css $opt" | tee -a $O/out.txt
    MIJ_VARIANTS=$V timeout -k 10 500 python3 tools/k1_variants.py run --css $css $opt --loop one-stream 2>&1 | tee -a $O/out.txt || exit 1
  done
done
