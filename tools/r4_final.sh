#!/bin/bash
# Round 4, final: the whole GPU suite, smoke(), then everything profiles/ quotes (tools/r4_refresh.sh) -- one GPU-box call.
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r4_fullsuite.sh || exit 1
grep -q " passed" gpurun_out/r4suite/pytest.txt || exit 1
grep -q "failed" gpurun_out/r4suite/pytest.txt && exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
bash tools/r4_refresh.sh
