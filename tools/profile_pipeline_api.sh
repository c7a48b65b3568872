#!/bin/bash
# rocprofv3 HIP-API trace + stats of tools/pipeline_host_rate.py (one rank over RCCL, four images in flight, 420 images):
# which runtime calls a steady-state step of sharded.DevicePipeline makes -- no stream synchronisation, no copy to the host.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_pipe_api
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --hip-trace --stats -d $out -o api --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pipeline_host_rate.py 5000 > $out/log.txt 2>&1
echo "profile rc=$?"
