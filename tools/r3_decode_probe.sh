#!/bin/bash
cd $GRAFT_REPO_ROOT
for sp in 0 65536; do for t in 0 64 96 128 192 256 384; do
  echo -n "MIJ_PAR_SPARSE=$sp MIJ_PAR_TAIL=$t: "; MIJ_PAR_SPARSE=$sp MIJ_PAR_TAIL=$t python tools/decode_fullsize.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('from host', [x[1] for x in d['wall_ms/device_ms per decode'][1:]], ' device-resident', [x[1] for x in d['same, file already in device memory']], d['psnr_db'])"
done; done
