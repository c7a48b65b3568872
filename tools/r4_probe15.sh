#!/bin/bash
# progressive no-DRI full-size decode with variant libraries (anchor spacing)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4px6; mkdir -p $O
for v in "$@"; do
  echo "== $v"
  L=""; [ "$v" != "tree" ] && L="MIJ_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/$v/libmijpeg.so"
  env $L MIJ_PX_DEBUG=1 timeout -k 10 300 python3 tools/decode_prog_nodri_fullsize.py 40000 4 2> $O/dbg_$v.txt | tail -1 | cut -c90-260,400-520
  grep "scan 5\|scan 9\|FELL" $O/dbg_$v.txt | tail -3 | cut -c1-140
done
