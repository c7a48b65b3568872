"""Full-size decode of this project's own PROGRESSIVE 4:2:2 q95 file (190 MB, DRI=104): device time per decode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvjpeg_imagecompressor_amd as mij
W, H = 8320, int(sys.argv[1]) if len(sys.argv) > 1 else 40000
dev = torch.device("cuda:0")
img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=True)
with mij.Encoder(W, H, 95, True, 1, progressive=True) as enc:
    enc.encode_device(img.data_ptr(), W * 3, "bgr")
    enc.result()
    jpg = enc.retrieve()
out = torch.empty_like(img)
with mij.Decoder() as dec:
    times = []
    for i in range(3):
        t0 = time.perf_counter()
        dec.decode_device(jpg, out.data_ptr(), W * 3, "bgr")
        ms = dec.sync()
        times.append((round((time.perf_counter() - t0) * 1e3, 2), round(ms, 2)))
diff = (out.to(torch.int16) - img.to(torch.int16)).float()
psnr = 10 * torch.log10(255.0 ** 2 / (diff * diff).mean()).item()
print(json.dumps({"jpeg_bytes": len(jpg), "wall_ms/device_ms per decode": times, "psnr_db": round(psnr, 3)}))
