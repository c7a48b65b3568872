#!/usr/bin/env python3
"""One small progressive file (no restart markers) through the decoder a few times; under `rocprofv3 --kernel-trace` the trace of the LAST decode says
where a sub-megapixel file's milliseconds go (launch count, busy time, span).  Usage: px_small_profile.py [W H Q SS]"""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O
W, H, Q, SS = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (416, 240, 95, 1)))
b = io.BytesIO()
Image.fromarray(O.synth_rgb(W, H)).save(b, "JPEG", progressive=True, optimize=True, quality=Q, subsampling=SS)
j = b.getvalue()
with mij.Decoder() as dec:
    for i in range(4):
        dec.decode_host(j, "rgb")
        print("decode %d: %.2f ms device, report %s" % (i, dec.last_ms(), dec.px_report()), flush=True)
