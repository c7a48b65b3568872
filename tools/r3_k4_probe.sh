#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-psnr "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transform %.4f tables %.4f entropy %.4f scan %.4f compact %.4f total %.4f crc %s' % (d['stage_ms']['transform'], d['stage_ms']['tables'], d['stage_ms']['entropy'], d['stage_ms']['scan'], d['stage_ms']['compact'], d['ms_per_step'], d['jpeg_crc32']))"; }
for i in 1 2; do
echo -n "persistent: "; run
echo -n "static:     "; MIJ_K4_STATIC=1 run
echo -n "persistent fixed: "; run --no-optimize
echo -n "static fixed:     "; MIJ_K4_STATIC=1 run --no-optimize
done
echo -n "persistent 444: "; run --css 444
echo -n "static 444:     "; MIJ_K4_STATIC=1 run --css 444
echo -n "persistent q99 (wide strips): "; run --quality 99
echo -n "static q99:     "; MIJ_K4_STATIC=1 run --quality 99
