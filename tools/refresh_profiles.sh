#!/bin/bash
# Regenerates everything under profiles/ that DESIGN.md quotes, in one GPU-box call; results land in gpurun_out/refresh/.
# Usage (from the repo root on the GPU box): bash tools/refresh_profiles.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
step() { echo "[$(date +%T)] $*" | tee -a $O/progress.txt; }
step "bench default";      timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
step "bench fixed";        timeout -k 10 300 python bench.py --no-optimize --no-cpu-baseline > $O/bench_fixed.json 2>> $O/bench_n1.err || exit 1
step "bench progressive";  timeout -k 10 300 python bench.py --progressive --no-cpu-baseline > $O/bench_prog.json 2>> $O/bench_n1.err || exit 1
step "bench two streams";  timeout -k 10 300 python bench.py --two-streams --no-cpu-baseline > $O/bench_two_streams.json 2>> $O/bench_n1.err || exit 1
step "bench 440 (IJG cpu leg)"; timeout -k 10 300 python bench.py --css 440 --steps 50 > $O/bench_440_cpu.json 2>> $O/bench_n1.err || exit 1
: > $O/table1.jsonl
for css in 444 422 440 420 411; do
  step "sampling $css"; timeout -k 10 300 python bench.py --css $css --no-cpu-baseline 2>> $O/bench_n1.err | tail -1 >> $O/table1.jsonl || exit 1
done
step "kernel trace";       bash tools/profile_bench.sh >> $O/progress.txt 2>&1 || exit 1
step "pmc fetch";          bash tools/pmc_pass.sh fetch FETCH_SIZE >> $O/progress.txt 2>&1 || exit 1
step "pmc write";          bash tools/pmc_pass.sh write WRITE_SIZE >> $O/progress.txt 2>&1 || exit 1
step "pmc sq a";           bash tools/pmc_pass.sh a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS >> $O/progress.txt 2>&1 || exit 1
step "pmc sq b";           bash tools/pmc_pass.sh b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS >> $O/progress.txt 2>&1 || exit 1
cd $R && python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_b > $O/hbm_traffic.json
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > $O/pmc_sq_summary.txt
step done
# second half: lines that quote the refreshed traffic file, progressive / decode kernel statistics, config 5
cp $O/hbm_traffic.json $R/profiles/r03_hbm_traffic.json
step "bench lines";        mkdir -p $O/lines && timeout -k 10 400 python bench.py > $O/lines/n1.json 2> $O/lines/err.txt && timeout -k 10 300 python bench.py --no-optimize --no-cpu-baseline > $O/lines/fixed.json 2>> $O/lines/err.txt && timeout -k 10 300 python bench.py --two-streams --no-cpu-baseline > $O/lines/two.json 2>> $O/lines/err.txt || exit 1
step "progressive stats";  bash tools/profile_prog1.sh >> $O/progress.txt 2>&1 || exit 1
step "decode";             timeout -k 10 300 python tools/decode_fullsize.py 2>/dev/null | tail -1 > $O/decode_fullsize.json || exit 1
step "decode stats";       bash tools/profile_decode.sh >> $O/progress.txt 2>&1 || exit 1
step "config 5";           timeout -k 10 300 python tools/secondary_fullsize.py 2>/dev/null | tail -1 > $O/secondary_fullsize.json || exit 1
step done2
