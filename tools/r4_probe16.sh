#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4px8; mkdir -p $O
for cfg in "X=1" "MIJ_PX_EVERY0=1" "MIJ_PX_EVERY0=2" "MIJ_PX_WSCALE=3" "MIJ_PX_EVERY0=1 MIJ_PX_WSCALE=3" "MIJ_PX_DEDUP=0 MIJ_PX_EVERY0=1 MIJ_PX_WSCALE=3"; do
  echo "== $cfg" | tee -a $O/out.txt
  for c in "1234x777 q90" "8320x2048 q75" "8320x2048 q90"; do
    env $cfg MIJ_PX_DEBUG=1 timeout -k 10 300 python3 tools/r4_px_test.py "$c" 2> $O/dbg.txt | grep "synth" | tee -a $O/out.txt
    grep "FELL" $O/dbg.txt | sort | uniq -c | cut -c1-150 | tee -a $O/out.txt
  done
done
