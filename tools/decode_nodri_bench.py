"""Decode of a libjpeg-turbo PROGRESSIVE file WITHOUT restart markers (what nvJPEG -- the reference's encoder -- and web /
camera files look like): wall + device time, pixels compared with libjpeg-turbo's own decoder (Pillow).
  python tools/decode_nodri_bench.py [height=4000] [width=8320]      (MIJ_DECODE_SCAN_LANES=1: the one-lane walk, for A/B)"""
import io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image, ImageFile
import nvjpeg_imagecompressor_amd as mij
ImageFile.MAXBLOCK = 1 << 30
Image.MAX_IMAGE_PIXELS = None
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8320
dev = torch.device("cuda:0")
img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=False)
torch.cuda.synchronize()
rgb = img.cpu().numpy()
b = io.BytesIO()
t0 = time.perf_counter()
Image.fromarray(rgb).save(b, "JPEG", quality=95, subsampling=1, progressive=True, optimize=True)
t_enc = time.perf_counter() - t0
jpg = b.getvalue()
assert b"\xff\xdd" not in jpg[:2000]
t0 = time.perf_counter()
want = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))
t_pil = time.perf_counter() - t0
out = torch.empty_like(img)
times = []
with mij.Decoder() as dec:
    for i in range(2):
        t0 = time.perf_counter()
        dec.decode_device(jpg, out.data_ptr(), W * 3, "rgb")
        ms = dec.sync()
        times.append((round((time.perf_counter() - t0) * 1e3, 1), round(ms, 1)))
same = bool(np.array_equal(out.cpu().numpy(), want))
print(json.dumps({"image": "%dx%d q95 4:2:2 progressive, no DRI (libjpeg-turbo via Pillow)" % (W, H), "jpeg_bytes": len(jpg),
                  "wall_ms/device_ms per decode": times, "pixels_identical_to_libjpeg_turbo": same,
                  "libjpeg_turbo_1core_decode_s": round(t_pil, 2), "libjpeg_turbo_1core_encode_s": round(t_enc, 2),
                  "scan_walk": "one lane" if os.environ.get("MIJ_DECODE_SCAN_LANES") else "wave"}))
