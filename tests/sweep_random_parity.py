"""Randomised GPU-vs-oracle sweep (a checker script, not collected by pytest; lives under tests/ because it uses the oracle): random sizes, samplings, qualities, restart intervals,
image statistics, baseline / fixed / progressive, encode bytes and decode pixels. Prints one line per failure."""
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # multiplies the size range (larger images: many subsequences / intervals)
bad = 0
t0 = time.time()
dec = mij.Decoder()
for it in range(N):
    W, H = int(rng.integers(1, 700 * SCALE)), int(rng.integers(1, 500 * SCALE))
    css = int(rng.integers(0, 6))
    q = int(rng.choice([1, 3, 10, 25, 50, 75, 90, 95, 100]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        img = O.synth_rgb(W, H)
    elif kind == 1:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    elif kind == 2:
        img = np.full((H, W, 3), rng.integers(0, 256, 3), np.uint8)
        img[rng.integers(0, H):, rng.integers(0, W):] = rng.integers(0, 256, 3)
    elif kind == 4:   # black / white noise: the largest blocks there are (entropy coder's roomy path, output-buffer growth)
        img = rng.integers(0, 2, (H, W, 3), dtype=np.uint8) * 255
    else:   # smooth gradient + sparse impulses: long zero runs with isolated coefficients
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
        for _ in range(20):
            img[rng.integers(0, H), rng.integers(0, W)] = rng.integers(0, 256, 3)
    mode = int(rng.integers(0, 3))   # 0 optimised, 1 fixed tables, 2 progressive
    ri = int(rng.choice([-1, 1, 2, 5, 17, 64]))
    fmt = str(rng.choice(["rgb", "bgr", "rgb_planar", "bgr_planar"]))      # input layout: same file whatever the layout
    src = img if fmt.startswith("rgb") else img[..., ::-1]
    src = np.ascontiguousarray(src.transpose(2, 0, 1)) if fmt.endswith("planar") else np.ascontiguousarray(src)
    try:
        with mij.Encoder(W, H, q, mode != 1, css, restart_interval=ri, progressive=(mode == 2)) as enc:
            r = enc.geometry["restart_interval"]
            got = enc.encode_host(src, fmt)
        want = O.encode_progressive(img, q, css, r) if mode == 2 else O.encode(img, q, css, mode != 1, r)
        ok = got == want
        if ok:
            pix = dec.decode_host(got, "rgb")
            ok = np.array_equal(pix, np.asarray(Image.open(io.BytesIO(got)).convert("RGB")))
        if ok and it % 3 == 0 and css < 3:
            # a third-party file of the same picture (libjpeg-turbo via Pillow): baseline / progressive, with / without DRI
            from PIL import ImageFile
            ImageFile.MAXBLOCK = 1 << 27
            kw = dict(quality=q, subsampling=css, progressive=bool(rng.integers(0, 2)), optimize=bool(rng.integers(0, 2)))
            if rng.integers(0, 2):
                kw["restart_marker_blocks"] = int(rng.integers(1, 50))
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", **kw)
            tp = b.getvalue()
            ok = np.array_equal(dec.decode_host(tp, "bgr")[..., ::-1], np.asarray(Image.open(io.BytesIO(tp)).convert("RGB")))
        if not ok:
            bad += 1
            print("FAIL", dict(W=W, H=H, css=css, q=q, kind=kind, mode=mode, ri=ri, fmt=fmt), flush=True)
    except Exception as e:
        bad += 1
        print("EXC", dict(W=W, H=H, css=css, q=q, kind=kind, mode=mode, ri=ri), repr(e)[:200], flush=True)
    if it % 50 == 49:
        print("... %d done, %d bad, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
dec.close()
print("sweep: %d cases, %d failures" % (N, bad))
sys.exit(1 if bad else 0)
