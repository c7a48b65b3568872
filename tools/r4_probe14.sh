#!/bin/bash
# progressive no-DRI: parity tests, then the full-size file over a few settings of the search (environment switches)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4px5; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_decode_generic.py -x -q -m gpu 2>&1 | tail -2 || exit 1
run() {
  echo "== $*"
  env "$@" MIJ_PX_DEBUG=1 timeout -k 10 300 python3 tools/decode_prog_nodri_fullsize.py 40000 4 nocheck 2> $O/dbg.txt | tail -1 | cut -c90-260
  grep "scan 5\|scan 9\|FELL" $O/dbg.txt | tail -3 | cut -c1-140
}
for cfg in "$@"; do run $cfg; done
