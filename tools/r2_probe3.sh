#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
for cfg in "1 0" "2 0" "3 0" "4 0" "2 1" "3 1" "4 1"; do
  set -- $cfg
  MIJ_PROG_STREAMS=$1 MIJ_PROG_ORDER=$2 timeout -k 10 200 python bench.py --progressive --steps 10 --warmup 2 --no-cpu-baseline --no-psnr > $O/p3_$1_$2.json 2> $O/p3_$1_$2.err
  python -c "import json;d=json.loads(open('$O/p3_$1_$2.json').read().strip().splitlines()[-1]);print('streams $1 lpt $2:', d['ms_per_step'], d['jpeg_crc32'])"
done
