#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p6; mkdir -p $O
python3 tools/r4_uncached_cost.py > $O/uncached_cost.txt 2>&1; echo "uncached rc=$?"; cat $O/uncached_cost.txt | tail -6
timeout -k 10 1500 python3 -m pytest tests/test_gpu_decode.py tests/test_gpu_sharded.py tests/test_gpu_bench.py tests/test_abi.py tests/test_gpu_cpp_facade.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
