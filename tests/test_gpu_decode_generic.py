"""Generic scan decoder (k_decode_scans.inc): progressive files -- the mode the reference's own encoder writes
(ImageCompressorImpl.cu:28) and its decoder reads back (.cu:335-366) -- greyscale, and files with restart markers in
every scan. The checker is libjpeg-turbo itself (Pillow): same coefficients => same IDCT / upsampling / colour
arithmetic as the baseline route, which is pinned pixel-for-pixel, so the comparison is exact."""
import io

import numpy as np
import pytest
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 26     # libjpeg cannot suspend while writing progressive / optimised output: give it the whole file

pytestmark = pytest.mark.gpu


def _pil_dec(j):
    return np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))


def _save(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def _images(oracle, W, H, seed):
    rng = np.random.default_rng(seed)
    yield oracle.synth_rgb(W, H)
    yield rng.integers(0, 256, (H, W, 3), dtype=np.uint8)                      # noise: long codes, many refinement bits
    flat = np.full((H, W, 3), 200, np.uint8)
    flat[H // 3:, W // 2:] = (10, 90, 250)                                     # mostly empty blocks: long EOB runs
    yield flat


@pytest.mark.parametrize("ss", [0, 1, 2])
@pytest.mark.parametrize("size", [(64, 64), (17, 33), (200, 136), (1, 1), (9, 250), (333, 77)])
def test_progressive_matches_libjpeg_turbo(mij, oracle, ss, size):
    W, H = size
    with mij.Decoder() as dec:
        for n, img in enumerate(_images(oracle, W, H, ss)):
            for q, opt in ((90, False), (35, True), (100, True)):
                jpg = _save(img, quality=q, subsampling=ss, progressive=True, optimize=opt)
                assert jpg[:2] == b"\xff\xd8" and b"\xff\xc2" in jpg        # really SOF2
                got = dec.decode_host(jpg, "rgb")
                want = _pil_dec(jpg)
                assert got.shape == want.shape
                assert np.array_equal(got, want), (n, q, opt, np.argwhere(got != want)[:4].tolist())


@pytest.mark.parametrize("rst", [1, 7])
def test_progressive_with_restart_markers(mij, oracle, rst):
    """DRI in a progressive file: every scan is cut into restart intervals (EOB runs end there), one lane each."""
    img = oracle.synth_rgb(176, 120)
    with mij.Decoder() as dec:
        for ss in (0, 2):
            jpg = _save(img, quality=85, subsampling=ss, progressive=True, restart_marker_blocks=rst)
            assert b"\xff\xdd" in jpg
            assert np.array_equal(dec.decode_host(jpg, "rgb"), _pil_dec(jpg))
            assert np.array_equal(dec.decode_host(jpg, "bgr"), _pil_dec(jpg)[..., ::-1])


@pytest.mark.parametrize("progressive", [False, True])
def test_greyscale(mij, oracle, progressive):
    g = oracle.synth_rgb(150, 93)[..., 1]
    with mij.Decoder() as dec:
        for q in (92, 40):
            b = io.BytesIO()
            Image.fromarray(g, "L").save(b, "JPEG", quality=q, progressive=progressive)
            jpg = b.getvalue()
            got = dec.decode_host(jpg, "rgb")
            assert np.array_equal(got, _pil_dec(jpg))
            info = mij.Decoder.info(jpg)
            assert (info["width"], info["height"]) == (150, 93)


def test_progressive_medium_image(mij, oracle):
    """A few megapixels without restart markers: each scan is one lane; checks it finishes and is exact."""
    img = oracle.synth_rgb(1536, 1024)
    jpg = _save(img, quality=90, subsampling=1, progressive=True, optimize=True)
    with mij.Decoder() as dec:
        assert np.array_equal(dec.decode_host(jpg, "rgb"), _pil_dec(jpg))
