"""Pins the CPU oracle DECODER (oracle/jpeg_oracle_dec.c) pixel-for-pixel against the stock decoder in this image
(libjpeg-turbo 3.1.4.1 through Pillow: islow IDCT, fancy upsampling, 16.16 colour conversion) and, for 4:4:0 / 4:1:1
files, against what IJG-written files decode to in libjpeg-turbo."""
import io
import os
import subprocess

import numpy as np
import pytest
from PIL import Image


def _pil_dec(j):
    return np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))


@pytest.mark.parametrize("css", range(6))
def test_decoder_matches_libjpeg_turbo(oracle, css):
    rng = np.random.default_rng(css)
    for (W, H) in [(1, 1), (2, 2), (3, 5), (5, 3), (8, 8), (9, 9), (17, 33), (33, 31), (64, 48), (100, 75), (129, 65), (250, 3)]:
        for img in (oracle.synth_rgb(W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8)):
            for q, ri in ((30, 0), (95, 3), (100, 0)):
                j = oracle.encode(img, q, css, True, ri)
                got = oracle.decode(j)
                assert np.array_equal(got, _pil_dec(j)), (W, H, q, ri)
                assert np.array_equal(oracle.decode(j, "bgr"), got[..., ::-1])
                inf = oracle.decode_info(j)
                assert (inf["width"], inf["height"], inf["restart_interval"]) == (W, H, ri)
                assert (inf["hs"], inf["vs"]) == oracle.CSS_FACTORS[css]


def test_decoder_on_third_party_files(oracle, tmp_path):
    """Files written by libjpeg-turbo itself and by IJG 9d (different marker order, different optimal tables)."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (75, 100, 3), dtype=np.uint8)
    for ss in (0, 1, 2):
        for kw in (dict(), dict(optimize=True), dict(restart_marker_blocks=4)):
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", quality=85, subsampling=ss, **kw)
            assert np.array_equal(oracle.decode(b.getvalue()), _pil_dec(b.getvalue()))
    harness = os.path.join(os.path.dirname(oracle.__file__), "ijg_harness")
    if os.path.exists(harness):
        raw = tmp_path / "in.raw"
        oracle.rgb_to_ycc(img).tofile(str(raw))
        for css in (3, 4, 5):
            hs, vs = oracle.CSS_FACTORS[css]
            out = tmp_path / "o.jpg"
            subprocess.check_call([harness, "enc", str(raw), "100", "75", "ycc", "80", str(hs), str(vs), "1", "3", str(out)])
            j = out.read_bytes()
            assert np.array_equal(oracle.decode(j), _pil_dec(j))


def test_entropy_decoder_inverts_entropy_coder(oracle):
    import ctypes as C
    L = oracle.lib()
    L.mjo_decode_coefficients.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    img = oracle.synth_rgb(136, 72)
    for css in range(6):
        coef = oracle.coefficients(img, 92, css)
        j = oracle.encode(img, 92, css, True, 5)
        back = np.empty_like(coef)
        buf = np.frombuffer(j, np.uint8)
        assert L.mjo_decode_coefficients(buf.ctypes.data, len(j), back.ctypes.data) == 0
        assert np.array_equal(back, coef)
