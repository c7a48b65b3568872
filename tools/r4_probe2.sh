#!/bin/bash
# round 4: the new bench line (driver's command), every loop at 20 and 200 steps, the bench tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p5; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { tail -20 $O/bench_driver_cmd.err; exit 1; }
for loop in overlap tables-ahead one-stream two-streams; do
  for steps in 20 200; do
    python3 bench.py --gpus 1 --steps $steps --warmup 5 --loop $loop --no-cpu-baseline --no-psnr > $O/bench_${loop}_$steps.json 2> $O/bench_${loop}_$steps.err || { tail -20 $O/bench_${loop}_$steps.err; exit 1; }
  done
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-optimize --no-cpu-baseline --no-psnr > $O/bench_fixed_20.json 2> $O/bench_fixed_20.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --settle-rounds 0 --no-cpu-baseline --no-psnr > $O/bench_nosettle_20.json 2> $O/bench_nosettle_20.err
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench.py -x -q > $O/pytest_bench.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_bench.txt
