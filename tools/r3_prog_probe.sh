#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --progressive --steps 15 --warmup 3 --no-cpu-baseline --no-psnr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms %.4f crc %s bytes %d' % (d['ms_per_step'], d['jpeg_crc32'], d['jpeg_bytes']))"; }
for st in 2 3 4; do for ord in 0 1; do echo -n "streams=$st order=$ord: "; MIJ_PROG_STREAMS=$st MIJ_PROG_ORDER=$ord run; done; done
