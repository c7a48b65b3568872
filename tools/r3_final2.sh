#!/bin/bash
# end of round 3, second half: the whole GPU suite on the library in the tree, then everything profiles/ quotes
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/final_gpu_tests.txt
[ $rc -eq 0 ] || exit 1
bash tools/r3_final.sh
