#!/bin/bash
# Everything profiles/ and DESIGN.md quote, on the library as it is, in GPU-box calls; results under gpurun_out/refresh/.
# Usage (repo root on the box):   bash tools/refresh_profiles.sh pmc      -> kernel traces + the four PMC passes + traffic / SQ summaries
#                                 bash tools/refresh_profiles.sh lines    -> the bench lines (quote the traffic / class files COMMITTED in profiles/)
#                                 bash tools/refresh_profiles.sh decode   -> progressive-decode cases and the thin full-size files
# Between `pmc` and `lines`, here (no GPU):  python tools/collect_profiles.py rNN  &&  python tools/valu_bound.py --traffic profiles/rNN_hbm_traffic.json > profiles/rNN_valu_classes.json
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
step() { echo "[$(date +%T)] $*" | tee -a $O/progress.txt; }
what=${1:-pmc}
if [ "$what" = pmc ]; then
  cd /tmp && export TMPDIR=/tmp
  step "kernel trace: the driver's command (no CPU legs, no extra sections)"
  rm -rf $R/gpurun_out/prof_bench; mkdir -p $R/gpurun_out/prof_bench
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-extra > $R/gpurun_out/prof_bench/log.txt 2>&1 || exit 1
  cd $R
  step "pmc fetch";  bash tools/pmc_pass.sh fetch FETCH_SIZE >> $O/progress.txt 2>&1 || exit 1
  step "pmc write";  bash tools/pmc_pass.sh write WRITE_SIZE >> $O/progress.txt 2>&1 || exit 1
  step "pmc sq a";   bash tools/pmc_pass.sh a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS >> $O/progress.txt 2>&1 || exit 1
  step "pmc sq b";   bash tools/pmc_pass.sh b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS >> $O/progress.txt 2>&1 || exit 1
  python3 tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_b gpurun_out/pmc_a > $O/hbm_traffic.json || exit 1
  python3 tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > $O/pmc_sq_summary.txt
elif [ "$what" = lines ]; then
  step "bench: the driver's command";  timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err || exit 1
  step "bench: 200 steps";             timeout -k 10 400 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_200.json 2>> $O/bench.err || exit 1
  step "bench: fixed tables";          timeout -k 10 300 python3 bench.py --steps 200 --no-optimize --no-cpu-baseline > $O/bench_fixed.json 2>> $O/bench.err || exit 1
  step "bench: progressive";           timeout -k 10 300 python3 bench.py --steps 50 --progressive --no-cpu-baseline > $O/bench_prog.json 2>> $O/bench.err || exit 1
  step "bench: progressive, 3 ranks on this one GPU (gloo rehearsal: correctness of the sharded path, not a rate)"
  MIJ_BENCH_ONE_DEVICE=1 timeout -k 10 500 python3 bench.py --gpus 3 --steps 5 --warmup 1 --progressive --no-cpu-baseline > $O/bench_prog_3ranks_one_device.json 2>> $O/bench.err || exit 1
else
  step "px cases";                     timeout -k 10 600 python3 tools/px_cases.py 2>/dev/null | grep -v "^\[px" > $O/px_cases.txt
  : > $O/px_thin.txt
  for qs in "95 1" "85 1" "90 2" "75 1"; do
    step "progressive no-DRI full size q/ss $qs"; timeout -k 10 900 python3 tools/decode_prog_nodri_fullsize.py 40000 3 check $qs 2>/dev/null | tail -1 >> $O/px_thin.txt
  done
fi
step "done $what"
