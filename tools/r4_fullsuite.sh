#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4suite; mkdir -p $O
( while true; do sleep 60; echo "[alive] $(date +%T) $(tail -c 200 $O/pytest.txt 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?
kill $HB
echo "pytest rc=$rc"; tail -15 $O/pytest.txt
