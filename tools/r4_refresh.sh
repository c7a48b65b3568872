#!/bin/bash
# Round 4: everything profiles/ and DESIGN.md quote, on the library as it is, in one GPU-box call; results under gpurun_out/r4refresh/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4refresh; mkdir -p $O
cd $R
step() { echo "[$(date +%T)] $*" | tee -a $O/progress.txt; }
step "bench: the driver's command";  timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err || exit 1
step "bench: 200 steps";             timeout -k 10 400 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_200.json 2>> $O/bench.err || exit 1
step "bench: one-stream loop";       timeout -k 10 300 python3 bench.py --steps 200 --loop one-stream --no-cpu-baseline --no-psnr > $O/bench_one_stream.json 2>> $O/bench.err || exit 1
step "bench: tables-ahead loop";     timeout -k 10 300 python3 bench.py --steps 200 --loop tables-ahead --no-cpu-baseline --no-psnr > $O/bench_tables_ahead.json 2>> $O/bench.err || exit 1
step "bench: fixed tables";          timeout -k 10 300 python3 bench.py --steps 200 --no-optimize --no-cpu-baseline > $O/bench_fixed.json 2>> $O/bench.err || exit 1
step "bench: progressive";           timeout -k 10 300 python3 bench.py --steps 50 --progressive --no-cpu-baseline > $O/bench_prog.json 2>> $O/bench.err || exit 1
: > $O/table1.jsonl
for css in 444 422 440 420 411; do
  step "sampling $css"; timeout -k 10 300 python3 bench.py --steps 100 --css $css --no-cpu-baseline 2>> $O/bench.err | tail -1 >> $O/table1.jsonl || exit 1
done
cd /tmp && export TMPDIR=/tmp
step "kernel trace: default bench (the driver's command, no CPU legs)"
rm -rf $R/gpurun_out/prof_bench; mkdir -p $R/gpurun_out/prof_bench
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > $R/gpurun_out/prof_bench/log.txt 2>&1 || exit 1
step "kernel trace: one-stream loop (every kernel alone: what roofline.avg_launch_ms is)"
rm -rf $R/gpurun_out/prof_bench_one; mkdir -p $R/gpurun_out/prof_bench_one
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench_one -o bench --output-format csv -- python3 $R/bench.py --steps 100 --loop one-stream --kernel-pass 1 --stage-a-pass 10 --no-cpu-baseline --no-psnr > $R/gpurun_out/prof_bench_one/log.txt 2>&1 || exit 1
cd $R
step "pmc fetch";  bash tools/pmc_pass.sh fetch FETCH_SIZE >> $O/progress.txt 2>&1 || exit 1
step "pmc write";  bash tools/pmc_pass.sh write WRITE_SIZE >> $O/progress.txt 2>&1 || exit 1
step "pmc sq a";   bash tools/pmc_pass.sh a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS >> $O/progress.txt 2>&1 || exit 1
step "pmc sq b";   bash tools/pmc_pass.sh b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS >> $O/progress.txt 2>&1 || exit 1
cd $R && python3 tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_b > $O/hbm_traffic.json || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > $O/pmc_sq_summary.txt
cp $O/hbm_traffic.json $R/profiles/r04_hbm_traffic.json
step "bench lines quoting the refreshed traffic file"
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench.err || exit 1
timeout -k 10 400 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_200.json 2>> $O/bench.err || exit 1
step "decode: own full-size file";   timeout -k 10 300 python3 tools/decode_fullsize.py 2>/dev/null | tail -1 > $O/decode_fullsize.json || exit 1
step "decode: progressive no-DRI full size"; timeout -k 10 600 python3 tools/decode_prog_nodri_fullsize.py 40000 5 2>/dev/null | tail -1 > $O/decode_prog_nodri_fullsize.json || exit 1
step "config 5";                     timeout -k 10 300 python3 tools/secondary_fullsize.py 2>/dev/null | tail -1 > $O/secondary_fullsize.json || exit 1
step "px cases";                     timeout -k 10 600 python3 tools/r4_px_test.py 2>/dev/null | grep -v "^\[px" > $O/px_cases.txt
step "progressive decode: kernel timeline"; TL_MIN_MS=0.25 bash tools/r4_probe10.sh > $O/px_timeline.txt 2>&1
step done
