#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 0 128 192 256; do echo "== TAIL=$t"; MIJ_PAR_TRACE=1 MIJ_PAR_TAIL=$t python tools/decode_fullsize.py 2>&1 | grep "\[par\]" | head -80 | awk '{print}' | tail -70 | head -40; done
