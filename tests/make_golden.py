"""Regenerates tests/golden/*.  Run from the repo root:  python tests/make_golden.py

The reference (OroChippw/Nvjpeg-ImageCompressor) ships no fixtures and cannot run here (closed-source nvJPEG + CUDA),
so the golden vectors are produced by the STOCK encoders that exist in this image on the SURVEY.md 8(d) synthetic input:
  * libjpeg-turbo 3.1.4.1 through Pillow  (4:4:4 / 4:2:2 / 4:2:0, fixed and optimised Huffman, with and without DRI)
  * IJG libjpeg 9d through oracle/ijg_harness (4:4:0 / 4:1:1 / 4:1:0 fixed Huffman, fed the oracle's YCbCr so the file
    depends only on downsample + FDCT + quantise + entropy coding)
Fixtures are data only: small JPEG files + a JSON index with lengths / CRC32 / PSNR.
"""
import io
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from PIL import Image, features  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def pil_encode(img, q, ss, opt, rst, progressive=False):
    b = io.BytesIO()
    kw = dict(quality=q, subsampling=ss, optimize=opt, progressive=progressive)
    if rst:
        kw["restart_marker_blocks"] = rst
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def ijg_encode(img, q, css, opt, rst):
    hs, vs = O.CSS_FACTORS[css]
    with tempfile.TemporaryDirectory() as td:
        raw, out = os.path.join(td, "i.raw"), os.path.join(td, "o.jpg")
        O.rgb_to_ycc(img).tofile(raw)
        H, W, _ = img.shape
        subprocess.check_call([os.path.join(ROOT, "oracle", "ijg_harness"), "enc", raw, str(W), str(H), "ycc", str(q),
                               str(hs), str(vs), str(int(opt)), str(rst), out])
        return open(out, "rb").read()


def main():
    O.build()
    os.makedirs(GOLD, exist_ok=True)
    index = {"generator": "tests/make_golden.py", "pillow": Image.__version__,
             "libjpeg_turbo": features.version_feature("libjpeg_turbo"), "ijg": "libjpeg 9d (/opt/conda)", "cases": []}
    small = O.synth_rgb(64, 48)
    odd = O.synth_rgb(37, 21)
    big = O.synth_rgb(512, 512)
    index["synthetic_crc32"] = {"64x48": "%08x" % zlib.crc32(small.tobytes()), "37x21": "%08x" % zlib.crc32(odd.tobytes()),
                                "512x512": "%08x" % zlib.crc32(big.tobytes())}
    for name, img in (("64x48", small), ("37x21", odd)):
        for css in (0, 1, 2):
            for opt in (False, True):
                for rst in (0, 2):
                    j = pil_encode(img, 95, css, opt, rst)
                    fn = "turbo_%s_css%d_%s_rst%d.jpg" % (name, css, "opt" if opt else "fix", rst)
                    open(os.path.join(GOLD, fn), "wb").write(j)
                    index["cases"].append(dict(file=fn, size=name, css=css, optimize=opt, restart=rst, quality=95,
                                               encoder="libjpeg-turbo", len=len(j), crc32="%08x" % zlib.crc32(j)))
            for rst in (0, 2):   # progressive (always optimised): the scan script and procedures of libjpeg, whole file
                j = pil_encode(img, 95, css, True, rst, progressive=True)
                fn = "turbo_%s_css%d_prog_rst%d.jpg" % (name, css, rst)
                open(os.path.join(GOLD, fn), "wb").write(j)
                index["cases"].append(dict(file=fn, size=name, css=css, optimize=True, restart=rst, quality=95, progressive=True,
                                           encoder="libjpeg-turbo", len=len(j), crc32="%08x" % zlib.crc32(j)))
        for css in (3, 4, 5):
            for rst in (0, 2):
                j = ijg_encode(img, 95, css, False, rst)
                fn = "ijg_%s_css%d_fix_rst%d.jpg" % (name, css, rst)
                open(os.path.join(GOLD, fn), "wb").write(j)
                index["cases"].append(dict(file=fn, size=name, css=css, optimize=False, restart=rst, quality=95,
                                           encoder="ijg9d-ycc", len=len(j), crc32="%08x" % zlib.crc32(j)))
    # 512x512 (BASELINE.json config 1 and its siblings): CRC + PSNR only
    for css in (0, 1, 2):
        for opt in (False, True):
            for rst in (0, 64):
                j = pil_encode(big, 95, css, opt, rst)
                dec = np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))
                index["cases"].append(dict(file=None, size="512x512", css=css, optimize=opt, restart=rst, quality=95,
                                           encoder="libjpeg-turbo", len=len(j), crc32="%08x" % zlib.crc32(j),
                                           psnr=round(O.psnr(big, dec), 3)))
    json.dump(index, open(os.path.join(GOLD, "index.json"), "w"), indent=1)
    print("wrote %d cases" % len(index["cases"]))


if __name__ == "__main__":
    main()
