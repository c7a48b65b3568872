#!/usr/bin/env python3
"""One-off randomized parity run on the GPU box: random sizes / samplings / qualities / restart intervals / pictures, every file
compared byte for byte with the oracle's, every other one also decoded by this library and compared with Pillow's pixels.
    python tools/fuzz_parity.py [cases] [seed]"""
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
dec = mij.Decoder()
for c in range(n_cases):
    W = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 700)), int(rng.integers(700, 2600))]))
    H = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 700)), int(rng.integers(700, 1500))]))
    css = int(rng.integers(0, 6))
    q = int(rng.choice([int(rng.integers(1, 101)), 95, 100, 75, 50]))
    opt = bool(rng.integers(0, 2))
    ri = int(rng.choice([-1, -1, int(rng.integers(1, 40)), int(rng.integers(40, 400)), 65535]))
    kind = int(rng.integers(0, 6))
    if kind == 0: img = O.synth_rgb(W, H)
    elif kind == 1: img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    elif kind == 2: img = np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)
    elif kind == 3: img = (rng.integers(0, 2, (H, W, 3), dtype=np.uint8) * 255)
    elif kind == 4:
        img = O.synth_rgb(W, H); img[:, W // 2:] = rng.integers(0, 256, (H, W - W // 2, 3), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:H, 0:W]; img = np.stack([(xx * 255 // max(1, W - 1)), (yy * 255 // max(1, H - 1)), ((xx + yy) & 255)], -1).astype(np.uint8)
    fmt = str(rng.choice(["rgb", "bgr", "rgb_planar"]))
    prog = bool(rng.integers(0, 5) == 0)
    src = img if fmt == "rgb" else (np.ascontiguousarray(img[..., ::-1]) if fmt == "bgr" else np.ascontiguousarray(img.transpose(2, 0, 1)))
    try:
        with mij.Encoder(W, H, q, True if prog else opt, css, restart_interval=ri, progressive=prog) as enc:
            rri = enc.geometry["restart_interval"]
            got = [enc.encode_host(src, fmt) for _ in range(int(rng.integers(1, 3)))]
        want = O.encode_progressive(img, q, css, rri) if prog else O.encode(img, q, css, opt, rri)
        ok = all(g == want for g in got)
        if ok and c % 2 == 0 and css != 5:
            px = dec.decode_host(want, "rgb")
            ref = np.asarray(Image.open(io.BytesIO(want)).convert("RGB"))
            ok = np.array_equal(px, ref)
    except Exception as ex:      # noqa: BLE001
        ok = False
        print("EXC", repr(ex))
    if not ok:
        bad += 1
        print("MISMATCH case", c, dict(W=W, H=H, css=css, q=q, opt=opt, ri=ri, kind=kind, fmt=fmt, prog=prog), flush=True)
    if c % 50 == 49:
        print("%d cases, %d bad, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("DONE %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
