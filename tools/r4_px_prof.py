#!/usr/bin/env python3
"""One progressive no-DRI file (synthetic, 8320 x ROWS, q95 4:2:2 by default) decoded a few times: device ms and which scans went parallel."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image, ImageFile
import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O
Image.MAX_IMAGE_PIXELS = None
ImageFile.MAXBLOCK = 1 << 28
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
q = int(sys.argv[2]) if len(sys.argv) > 2 else 95
ss = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
img = O.synth_rgb(8320, 40000, y0=0, rows=rows)
b = io.BytesIO()
Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=ss, progressive=True, optimize=True)
j = b.getvalue()
ref = np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))
with mij.Decoder() as dec:
    for r in range(reps):
        got = dec.decode_host(j, "rgb")
        print("rep %d: device %.2f ms, parallel %s, identical %s" % (r, dec.last_ms(), dec.px_report(), np.array_equal(got, ref)), flush=True)
print("file %d bytes, %d x %d" % (len(j), 8320, rows))
