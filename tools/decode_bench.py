"""Decode timings (device time from mij_decode_sync): own baseline+DRI files vs third-party progressive files."""
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image, ImageFile

import torch

import nvjpeg_imagecompressor_amd as mij

ImageFile.MAXBLOCK = 1 << 28
Image.MAX_IMAGE_PIXELS = None


def main():
    W, H = 8320, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    d = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    mij.synth_image_device(d.data_ptr(), W, 0, H, W * 3, bgr=False)     # the library's device generator, RGB
    torch.cuda.synchronize()
    img = d.cpu().numpy()
    del d
    out = {}
    with mij.Encoder(W, H, 95, True, 1) as enc:
        own = enc.encode_host(img, "rgb")
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=95, subsampling=1, progressive=True, optimize=True)
    prog = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=95, subsampling=1, optimize=True)
    base = b.getvalue()
    with mij.Decoder() as dec:
        for name, j in (("own_baseline_dri", own), ("turbo_baseline_nodri", base), ("turbo_progressive", prog)):
            dec.decode_host(j, "bgr")
            t0 = time.perf_counter()
            got = dec.decode_host(j, "bgr")
            dt = time.perf_counter() - t0
            ok = bool(np.array_equal(got[..., ::-1], np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))))
            out[name] = {"bytes": len(j), "host_to_host_ms": round(dt * 1e3, 2), "Mpixels/s": round(W * H / 1e6 / dt, 1), "pixel_exact_vs_libjpeg_turbo": ok}
    print(json.dumps({"image": "%dx%d q95 4:2:2" % (W, H), **out}))


if __name__ == "__main__":
    main()
