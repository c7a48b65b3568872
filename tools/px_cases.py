#!/usr/bin/env python3
"""Progressive files WITHOUT restart markers through the parallel decoder (k_decode_prog.inc) against Pillow, pixel for pixel:
samplings, sizes, qualities, smooth and noisy content; reports which scans went parallel and the device time."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O

Image.MAX_IMAGE_PIXELS = None
from PIL import ImageFile
ImageFile.MAXBLOCK = 1 << 26          # progressive + optimised output of noise needs more than Pillow's default buffer
rng = np.random.default_rng(5)
cases = []
for (w, h) in ((416, 240), (1040, 512), (8320, 2048), (1234, 777)):
    base = O.synth_rgb(w, h)
    for q, ss in ((95, 1), (95, 0), (90, 2), (75, 1), (98, 2)):
        cases.append(("synth %dx%d q%d ss%d" % (w, h, q, ss), base, dict(quality=q, subsampling=ss)))
noise = rng.integers(0, 256, (600, 800, 3), dtype=np.uint8)
cases.append(("noise q97", noise, dict(quality=97, subsampling=0)))
cases.append(("noise q50", noise, dict(quality=50, subsampling=2)))
yy, xx = np.mgrid[0:900, 0:1400]
smooth = np.stack([(xx * 255 // 1400), (yy * 255 // 900), ((xx + yy) * 255 // 2300)], -1).astype(np.uint8)
cases.append(("smooth gradient q90", smooth, dict(quality=90, subsampling=2)))
cases.append(("grey synth", O.synth_rgb(1040, 512)[..., 1], dict(quality=92)))
bad = 0
only = sys.argv[1] if len(sys.argv) > 1 else None
with mij.Decoder() as dec:
    for name, img, kw in cases:
        if only and only not in name:
            continue
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", progressive=True, optimize=True, **kw)
        j = b.getvalue()
        ref = np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))
        dec.decode_host(j, "rgb")            # first call: the workspaces are allocated
        t0 = time.perf_counter()
        got = dec.decode_host(j, "rgb")
        dt = time.perf_counter() - t0
        tried, par = dec.px_report()
        same = np.array_equal(got, ref)
        bad += 0 if same else 1
        print("%-28s %8d B  %s  scans parallel %d / %d  device %.2f ms (host call %.1f ms)" % (name, len(j), "ok " if same else "MISMATCH", par, tried, dec.last_ms(), dt * 1e3), flush=True)
        if not same:
            d = np.argwhere((got != ref).any(-1))
            print("   first mismatch at", d[0], "count", len(d))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
