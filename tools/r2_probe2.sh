#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_progressive.py -x -q -p no:cacheprovider > $O/p2_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/p2_tests.log; tail -25 $O/p2_tests.log
timeout -k 10 300 python bench.py --progressive --steps 10 --warmup 2 --no-cpu-baseline --no-psnr > $O/p2_bench_prog.json 2> $O/p2_bench_prog.err; echo "prog bench rc=$?"; tail -c 1500 $O/p2_bench_prog.json; tail -5 $O/p2_bench_prog.err
MIJ_PROG_SERIAL=1 timeout -k 10 300 python bench.py --progressive --steps 5 --warmup 1 --no-cpu-baseline --no-psnr > $O/p2_bench_prog_serial.json 2> $O/p2_bench_prog_serial.err; echo "serial prog bench rc=$?"; tail -c 600 $O/p2_bench_prog_serial.json
