#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p9; mkdir -p $O
MIJ_PX_DEBUG=1 timeout -k 10 900 python3 tools/r4_px_test.py > $O/px_test.txt 2>&1; echo "rc=$?"; grep -v "^\[px\]" $O/px_test.txt | tail -26; grep -c "FELL BACK" $O/px_test.txt
MIJ_PX_DEBUG=1 timeout -k 10 900 python3 tools/decode_prog_nodri_fullsize.py 40000 2 > $O/full.txt 2>&1; echo "full rc=$?"; grep -v "^\[px\] scan [0-4]" $O/full.txt | tail -30
