#!/bin/bash
# the fused inverse transform + colour conversion: decoder tests with it and without, then full-size decode and config 5 both ways
cd $GRAFT_REPO_ROOT
for f in 1 0; do
  MIJ_FUSED_IDCT=$f python -m pytest tests/test_gpu_decode.py tests/test_gpu_decode_generic.py tests/test_gpu_cpp_facade.py -x -q -m gpu > gpurun_out/fused_tests_$f.txt 2>&1 || { echo "MIJ_FUSED_IDCT=$f"; tail -30 gpurun_out/fused_tests_$f.txt; exit 1; }
  echo "MIJ_FUSED_IDCT=$f: $(tail -1 gpurun_out/fused_tests_$f.txt)"
done
for f in 0 1; do
  echo "MIJ_FUSED_IDCT=$f decode:    $(MIJ_FUSED_IDCT=$f python tools/decode_fullsize.py 2>/dev/null | tail -1)"
  echo "MIJ_FUSED_IDCT=$f secondary: $(MIJ_FUSED_IDCT=$f python tools/secondary_fullsize.py 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/fused_step.txt
