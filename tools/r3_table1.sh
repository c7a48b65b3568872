#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O; cd $R
: > $O/table1.jsonl
for css in 444 422 440 420 411; do timeout -k 10 300 python bench.py --css $css --no-cpu-baseline 2>> $O/table1.err | tail -1 >> $O/table1.jsonl || exit 1; done
mkdir -p $O/lines; timeout -k 10 300 python bench.py --no-optimize --no-cpu-baseline > $O/lines/fixed.json 2>> $O/table1.err || exit 1
python - <<PY
import json
for l in open("$O/table1.jsonl"):
    x=json.loads(l); print(x['config']['workload'][68:74], x['ms_per_step'], x['value'], x['ratio'], x['psnr_db'], x['stage_ms']['transform'], x['stage_ms']['entropy'], x['jpeg_crc32'])
PY
