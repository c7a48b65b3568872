#!/bin/bash
# does the stage-A pass's pair of fixed-table encoders (allocated before the loop) cost the timed loop anything? same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
  for sa in 10 0; do
    timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --stage-a-pass $sa 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stage-a-pass $sa:', d['ms_per_step'], d['value'], d['clock']['settle_steps'], [c['counter_MHz'] for c in d['clock']['probes']][-4:])"
  done
done
