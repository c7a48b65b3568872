#!/bin/bash
# round 3: the decoder's symbol loop (real prefetch, per-block table choice) and its look-ahead / long-code variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_decode.py -x -q -m gpu > gpurun_out/decode_loop_tests.txt 2>&1 || { tail -30 gpurun_out/decode_loop_tests.txt; exit 1; }
tail -2 gpurun_out/decode_loop_tests.txt
MIJ_VARIANTS=dec_default,dec_lim,dec_all256,dec_lb10_128,dec_lb10_256,dec_lb11_256,dec_lb11_512,dec_lb12_512 MIJ_PAR_TRACE=1 python tools/decode_variants.py run 2>&1 | tee gpurun_out/decode_loop_variants.txt
