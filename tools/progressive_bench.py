"""Full-size progressive encode (8320x40000 q95 4:2:2): time, size vs baseline, decode check of a strip with Pillow."""
import io
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nvjpeg_imagecompressor_amd as mij

W, H = 8320, int(sys.argv[1]) if len(sys.argv) > 1 else 40000
dev = torch.device("cuda:0")
img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=True)
torch.cuda.synchronize()
out = {}
for name, prog in (("baseline", False), ("progressive", True)):
    with mij.Encoder(W, H, 95, True, 1, progressive=prog) as enc:
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enc.encode_device(img.data_ptr(), W * 3, "bgr")
            r = enc.result()
            times.append((time.perf_counter() - t0) * 1e3)
        jpg = enc.retrieve()
        out[name] = {"ms": round(min(times), 2), "bytes": len(jpg), "crc32": "%08x" % zlib.crc32(jpg)}
out["progressive_vs_baseline_size"] = round(out["progressive"]["bytes"] / out["baseline"]["bytes"], 4)
print(json.dumps({"image": "%dx%d q95 4:2:2" % (W, H), **out}))
