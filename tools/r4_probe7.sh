#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p7; mkdir -p $O
MIJ_PX_DUMP=1 MIJ_PX_DEBUG=1 timeout -k 10 600 python3 tools/r4_px_test.py "1234x777 q95 ss1" > $O/px_dump.txt 2>&1; echo "rc=$?"; grep "px-dump" $O/px_dump.txt | head -60
