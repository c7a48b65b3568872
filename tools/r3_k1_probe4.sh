#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-psnr "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transform %.4f entropy %.4f total %.4f' % (d['stage_ms']['transform'], d['stage_ms']['entropy'], d['ms_per_step']))"; }
for css in 420 411 444 440; do for n in 4 6 8 10 12 16; do echo -n "css=$css wg_per_cu=$n: "; MIJ_K1_WG_PER_CU=$n run --css $css; done; done
