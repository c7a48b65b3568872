#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p11; mkdir -p $O
MIJ_PROG_PARALLEL=0 timeout -k 10 900 python3 tools/r4_px_test.py "416x240" > $O/px_off_small.txt 2>&1
MIJ_PROG_PARALLEL=0 timeout -k 10 900 python3 tools/r4_px_test.py "1040x512" >> $O/px_off_small.txt 2>&1
MIJ_PROG_PARALLEL=0 timeout -k 10 900 python3 tools/r4_px_test.py "1234x777" >> $O/px_off_small.txt 2>&1
grep "^synth" $O/px_off_small.txt
