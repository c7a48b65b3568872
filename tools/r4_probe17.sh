#!/bin/bash
# progressive no-DRI: parity tests, the content cases, the full-size file -- one call
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4px9; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_decode_generic.py -x -q -m gpu 2>&1 | tail -2 || exit 1
MIJ_PX_DEBUG=1 timeout -k 10 600 python3 tools/r4_px_test.py 2> $O/cases_dbg.txt | grep -v "^\[px" > $O/cases.txt; cat $O/cases.txt
grep "candidate lists\|FELL" $O/cases_dbg.txt | sort | uniq -c | sort -rn | head -30
MIJ_PX_DEBUG=1 timeout -k 10 300 python3 tools/decode_prog_nodri_fullsize.py 40000 5 2> $O/dbg.txt | tail -1 > $O/full.json; cut -c90-330 $O/full.json
grep "scan 5\|scan 9\|FELL\|candidate" $O/dbg.txt | tail -4 | cut -c1-140
