#!/bin/bash
# decoder tests, then the full-size decode and its per-kernel times with the library in the tree
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decode.py tests/test_gpu_decode_generic.py -x -q -m gpu > gpurun_out/decode_step_tests.txt 2>&1 || { tail -30 gpurun_out/decode_step_tests.txt; exit 1; }
tail -1 gpurun_out/decode_step_tests.txt
MIJ_PAR_TRACE=1 python tools/decode_fullsize.py 2>gpurun_out/decode_step_err.txt | tail -1
grep "^\[par\]" gpurun_out/decode_step_err.txt | tail -3
bash tools/r3_decode_prof2.sh > gpurun_out/decode_step_prof.txt 2>&1; grep -E "k_par|k_idct|k_upsample|k_clean|rc=" gpurun_out/decode_step_prof.txt | head -14
