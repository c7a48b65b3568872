#!/bin/bash
# the two headline lines + the kernel trace of the driver's command, with the bench.py of the tree (library unchanged)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4refresh; mkdir -p $O; cd $R
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err || exit 1
timeout -k 10 400 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_200.json 2>> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench; mkdir -p $R/gpurun_out/prof_bench
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > $R/gpurun_out/prof_bench/log.txt 2>&1 || exit 1
cd $R; python3 - <<'PY'
import json
for f in ("bench_driver_cmd", "bench_200"):
    d = json.loads(open("gpurun_out/r4refresh/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["stage_A_alone"]["frac"], d["clock"]["settle_steps"], [c["counter_MHz"] for c in d["clock"]["probes"]])
PY
