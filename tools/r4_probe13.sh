#!/bin/bash
# progressive no-DRI: parity tests, the full-size file (5 decodes, pixel check), then the kernel timeline of two decodes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4px4; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_decode_generic.py -x -q -m gpu 2>&1 | tail -2 || exit 1
MIJ_PX_COUNTS=1 timeout -k 10 300 python3 tools/decode_prog_nodri_fullsize.py 40000 1 nocheck > $O/out_counts.txt 2> $O/counts.txt || exit 1
timeout -k 10 300 python3 tools/decode_prog_nodri_fullsize.py 40000 5 > $O/out.txt 2>/dev/null || exit 1
tail -1 $O/out.txt
bash tools/r4_probe10.sh > $O/timeline.txt 2>&1
