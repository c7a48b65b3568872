"""Full-size (8320x40000) decode of this project's own 4:2:2 q95 file, device-resident output: device time per decode."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nvjpeg_imagecompressor_amd as mij

W, H = 8320, 40000
dev = torch.device("cuda:0")
img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=True)
with mij.Encoder(W, H, 95, True, 1) as enc:
    enc.encode_device(img.data_ptr(), W * 3, "bgr")
    r = enc.result()
    jpg = enc.retrieve()
    d_jpg = torch.empty(len(jpg), dtype=torch.uint8, device=dev)          # a device-resident copy of the file
    from nvjpeg_imagecompressor_amd import sharded
    d_jpg.copy_(sharded.device_bytes(torch, r["d_buffer"] + r["header_offset"], r["file_bytes"], dev))
out = torch.empty_like(img)
with mij.Decoder() as dec:
    times, times_dev = [], []
    for i in range(4):
        t0 = time.perf_counter()
        dec.decode_device(jpg, out.data_ptr(), W * 3, "bgr")
        ms = dec.sync()
        times.append((round((time.perf_counter() - t0) * 1e3, 2), round(ms, 2)))
    for i in range(3):
        out.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.decode_device_ptr(d_jpg.data_ptr(), len(jpg), out.data_ptr(), W * 3, "bgr")
        ms = dec.sync()
        times_dev.append((round((time.perf_counter() - t0) * 1e3, 2), round(ms, 2)))
diff = (out.to(torch.int16) - img.to(torch.int16)).float()
psnr = 10 * torch.log10(255.0 ** 2 / (diff * diff).mean()).item()
print(json.dumps({"jpeg_bytes": len(jpg), "wall_ms/device_ms per decode": times, "same, file already in device memory": times_dev,
                  "psnr_db": round(psnr, 3)}))
