#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p3; mkdir -p $O
python3 tools/r4_clock_transient.py > $O/transient.txt 2>&1; echo rc=$?; tail -40 $O/transient.txt
