#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-psnr "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transform %.4f entropy %.4f total %.4f crc %s' % (d['stage_ms']['transform'], d['stage_ms']['entropy'], d['ms_per_step'], d['jpeg_crc32']))"; }
for opt in "" "--no-optimize"; do
  echo -n "ticket(4/CU) $opt: "; run $opt
  for n in 3 5 8; do echo -n "ticket wg_per_cu=$n $opt: "; MIJ_K1_WG_PER_CU=$n run $opt; done
  echo -n "static 6/CU $opt: "; MIJ_K1_STATIC=1 run $opt
  echo -n "static 16/CU $opt: "; MIJ_K1_STATIC=1 MIJ_K1_WG_PER_CU=16 run $opt
done
