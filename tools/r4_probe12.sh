#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p12; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_encode.py tests/test_gpu_tables.py tests/test_gpu_progressive.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > $O/bench_20.json 2> $O/bench_20.err
python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-psnr > $O/bench_200.json 2> $O/bench_200.err
python3 bench.py --gpus 1 --steps 200 --warmup 5 --loop one-stream --no-cpu-baseline --no-psnr > $O/bench_200_one.json 2> $O/bench_200_one.err
for f in bench_20 bench_200 bench_200_one; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['stage_A_alone']['frac'], d['golden_match'], [int(p['counter_MHz']) for p in d['clock']['probes']])
"; done
