#!/bin/bash
# The whole GPU suite in ONE process on the GPU box, with a heartbeat line a minute (gpurun takes a silent command for hung), then smoke().
# Usage (repo root, on the box): bash tools/gpu_suite.sh [pytest args]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/suite; mkdir -p $O
( while true; do sleep 60; echo "[alive] $(date +%T) $(tail -c 200 $O/pytest.txt 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu "$@" > $O/pytest.txt 2>&1; rc=$?
kill $HB
echo "pytest rc=$rc"; tail -15 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
