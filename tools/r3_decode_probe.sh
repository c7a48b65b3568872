#!/bin/bash
# full-size decode against the speculative tail (MIJ_PAR_TAIL bytes) and the sparse-pass threshold (MIJ_PAR_SPARSE items)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for sp in 1024 8192 65536; do for t in 0 128 256 384 512; do
  echo -n "MIJ_PAR_SPARSE=$sp MIJ_PAR_TAIL=$t: "; MIJ_PAR_TRACE=1 MIJ_PAR_SPARSE=$sp MIJ_PAR_TAIL=$t python tools/decode_fullsize.py 2>gpurun_out/probe_err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('from host', [x[1] for x in d['wall_ms/device_ms per decode'][1:]], ' device-resident', [x[1] for x in d['same, file already in device memory']], d['psnr_db'], end=' ')"
  grep "^\[par\]" gpurun_out/probe_err.txt | tail -8 | awk '{printf "%s%s ", $3=="(sparse):"?"s":"d", $4}' ; echo
done; done 2>&1 | tee gpurun_out/decode_tail_probe.txt
