#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-psnr "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transform %.4f total %.4f' % (d['stage_ms']['transform'], d['ms_per_step']))"; }
for n in 16 20 24 32 40 48 64 80; do echo -n "nostats wg_per_cu=$n: "; MIJ_K1_WG_PER_CU=$n run --no-optimize; done
for n in 14 16 18 20; do echo -n "stats wg_per_cu=$n: "; MIJ_K1_WG_PER_CU=$n run; done
