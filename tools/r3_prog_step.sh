#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_progressive.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for i in 1 2; do python bench.py --progressive --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('progressive', d['ms_per_step'], d['value'], d.get('jpeg_crc32'))"; done
bash tools/profile_prog1.sh > /dev/null 2>&1; python3 - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/prof_prog1/**/*kernel_stats.csv',recursive=True)[0])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    if 'prog2' in r['Name'] or 'transform' in r['Name']: print("%-80s calls %4s avg %8.1f us" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
