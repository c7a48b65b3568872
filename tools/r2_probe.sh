#!/bin/bash
# round-2 probe on the GPU box: fused entropy coder sanity + A/B, microbenchmarks, K1 variants. Output under gpurun_out/.
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fullsize.py tests/test_gpu_tables.py -x -q -p no:cacheprovider > $O/p_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/p_tests.log; tail -3 $O/p_tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-psnr > $O/p_bench_fused_$i.json 2> $O/p_bench_fused_$i.err; echo "fused rc=$?"
MIJ_FUSE=1 timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-psnr > $O/p_bench_optfused_$i.json 2> $O/p_bench_optfused_$i.err; echo "opt-in fused rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/p_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["stage_ms"], d["jpeg_crc32"])
    except Exception as e: print(f, "FAILED", e)
PY
timeout -k 10 120 tools/copy_rate > $O/p_copy_rate.txt 2>&1; cat $O/p_copy_rate.txt
timeout -k 10 300 tools/issue_mix 2 3 4 8 > $O/p_issue_mix.txt 2>&1; cat $O/p_issue_mix.txt
if [ -d build/variants ]; then timeout -k 10 600 python tools/k1_variants.py run > $O/p_k1_variants.txt 2>&1; cat $O/p_k1_variants.txt; fi
