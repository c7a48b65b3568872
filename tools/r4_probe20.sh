#!/bin/bash
# sweeps with the fitted estimate: parity tests, content cases, the dense full-size file, then the thin full-size files
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r4_probe17.sh 2>&1 | grep -v "candidate lists" || exit 1
O=gpurun_out/r4px11; mkdir -p $O
for cfg in "90 2" "75 1" "85 1"; do
  MIJ_PX_DEBUG=1 timeout -k 10 400 python3 tools/decode_prog_nodri_fullsize.py 40000 3 check $cfg 2> $O/dbg.txt | tail -1 | cut -c1-330
  grep "FELL\|adopted" $O/dbg.txt | tail -5 | cut -c1-150
done
