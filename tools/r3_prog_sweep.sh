#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 2 3 4; do for ord in 1 0; do
  echo -n "MIJ_PROG_STREAMS=$st MIJ_PROG_ORDER=$ord: "
  MIJ_PROG_STREAMS=$st MIJ_PROG_ORDER=$ord python bench.py --progressive --no-cpu-baseline --no-psnr 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('jpeg_crc32'))"
done; done 2>&1 | tee gpurun_out/prog_sweep.txt
