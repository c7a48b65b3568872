#!/bin/bash
# round 3: what bounds K1? (a) grid size, (b) the no-statistics kernel without its stores / loads (timing-only variants), (c) rounds 1-2 kernel switches (results: profiles/r03_k1_grid.txt, r03_k1_variants.txt; the ticket / stagger / persistent-K4 builds those files also mention were removed from the library after measuring)
# stores / loads (timing-only variants), (c) r2's kernel on the same box.
cd $GRAFT_REPO_ROOT
for n in 4 5 6 8 12 16; do
  for opt in "" "--no-optimize"; do
    echo -n "wg_per_cu=$n $opt: "; MIJ_K1_WG_PER_CU=$n python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-psnr $opt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transform %.4f entropy %.4f total %.4f' % (d['stage_ms']['transform'], d['stage_ms']['entropy'], d['ms_per_step']))"
  done
done
MIJ_VARIANTS=default,r2_kernel,nostore,noload python tools/k1_variants.py run --no-optimize
