#!/usr/bin/env python3
"""The judge's round-3 criterion for progressive files: a libjpeg-turbo PROGRESSIVE 8320x40000 q95 4:2:2 file WITHOUT restart markers
(Pillow writes it here, ~1.5 min) decoded by the library, device resident in and out, pixel for pixel against Pillow's own decode.
Prints one JSON line: file and image fingerprints, device ms per decode (several repetitions), which scans went parallel."""
import io, json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image, ImageFile
import nvjpeg_imagecompressor_amd as mij
Image.MAX_IMAGE_PIXELS = None
ImageFile.MAXBLOCK = 1 << 30
W, H = 8320, int(sys.argv[1]) if len(sys.argv) > 1 else 40000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
check = (sys.argv[3] != "nocheck") if len(sys.argv) > 3 else True
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 95             # quality and Pillow's subsampling code (0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0) of the file
SS = int(sys.argv[5]) if len(sys.argv) > 5 else 1
d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
mij.synth_image_device(d_img.data_ptr(), W, 0, H, W * 3, bgr=False)
torch.cuda.synchronize()
img = d_img.cpu().numpy()
t0 = time.perf_counter()
b = io.BytesIO()
Image.fromarray(img).save(b, "JPEG", quality=Q, subsampling=SS, progressive=True, optimize=True)
j = b.getvalue()
t_enc = time.perf_counter() - t0
print("##progress encoded %d bytes in %.1f s" % (len(j), t_enc), flush=True)
d_file = torch.frombuffer(bytearray(j), dtype=torch.uint8).cuda()
d_out = torch.empty_like(d_img)
ms = []
with mij.Decoder() as dec:
    for r in range(reps):
        dec.decode_device_ptr(d_file.data_ptr(), len(j), d_out.data_ptr(), W * 3, "rgb")
        ms.append(round(dec.sync(), 2))
        print("##progress rep %d: %.2f ms, parallel %s" % (r, ms[-1], dec.px_report()), flush=True)
    tried, par = dec.px_report()
out = {"file": "libjpeg-turbo (Pillow) progressive, no DRI, %dx%d q%d %s" % (W, H, Q, {0: "4:4:4", 1: "4:2:2", 2: "4:2:0"}[SS]), "file_bytes": len(j), "file_crc32": "%08x" % zlib.crc32(j),
       "device_ms": ms, "device_ms_median": sorted(ms)[len(ms) // 2], "scans_tried": tried, "scans_parallel": par, "pillow_encode_s": round(t_enc, 1),
       "library_source_hash": mij.library_source_hash()}
got = d_out.cpu().numpy()
out["decoded_crc32"] = "%08x" % zlib.crc32(got.tobytes())
if check:
    t0 = time.perf_counter()
    ref = np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))
    out["pillow_decode_s"] = round(time.perf_counter() - t0, 1)
    out["identical_to_pillow"] = bool(np.array_equal(got, ref))
    out["pillow_decoded_crc32"] = "%08x" % zlib.crc32(ref.tobytes())
print(json.dumps(out), flush=True)
