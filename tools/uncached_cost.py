#!/usr/bin/env python3
"""What does a device-uncached output buffer (mij_encoder_reserve_output, the sharded path) cost the kernels that write and read
it? Whole image and a 1/8 strip, stage times with the plain and the uncached buffer, and a device-to-device copy out of it."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvjpeg_imagecompressor_amd as mij

W = 8320
torch.cuda.set_device(0)
s = torch.cuda.current_stream().cuda_stream
for H in (40000, 5000):
    img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=True, stream=s)
    for reserve in (False, True):
        with mij.Encoder(W, H, 95, True, 1) as enc:
            if reserve:
                enc.reserve_output(enc.output_buffer()[2] * 8 + 1)
            enc.enable_timing(True)
            acc = {}
            for i in range(12):
                enc.encode_device(img.data_ptr(), W * 3, "bgr", 0, s)
                r = enc.result()
                if i >= 2:
                    for k, v in enc.stage_times().items():
                        acc[k] = acc.get(k, 0.0) + v / 10
            t = mij.sharded.device_bytes(torch, r["d_buffer"] + r["header_offset"], r["file_bytes"], img.device) if hasattr(mij, "sharded") else None
            from nvjpeg_imagecompressor_amd import sharded
            t = sharded.device_bytes(torch, r["d_buffer"] + r["header_offset"], r["file_bytes"], img.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c = t.clone()
            e0.record()
            for _ in range(10):
                c.copy_(t)
            e1.record(); e1.synchronize()
            print(json.dumps({"rows": H, "uncached": enc.output_is_uncached(), "stage_ms": {k: round(v, 4) for k, v in acc.items()},
                              "file_MB": round(r["file_bytes"] / 1e6, 1), "copy_out_ms": round(e0.elapsed_time(e1) / 10, 4)}), flush=True)
