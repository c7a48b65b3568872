#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-psnr "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transform %.4f entropy %.4f total %.4f crc %s' % (d['stage_ms']['transform'], d['stage_ms']['entropy'], d['ms_per_step'], d['jpeg_crc32']))"; }
export MIJ_K1_STATIC=1
for opt in "" "--no-optimize"; do
  for n in 4 6 16; do for st in 0 2 3 4 6; do
    echo -n "static wg_per_cu=$n stagger=$st $opt: "; MIJ_K1_WG_PER_CU=$n MIJ_K1_STAGGER=$st run $opt
  done; done
done
for n in 10 12 16 20 24 32 48; do echo -n "static wg_per_cu=$n stats: "; MIJ_K1_WG_PER_CU=$n run; done
for n in 7 8 9 10 12; do echo -n "static wg_per_cu=$n nostats: "; MIJ_K1_WG_PER_CU=$n run --no-optimize; done
