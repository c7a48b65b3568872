"""BASELINE.json's full-size configurations on the GPU: bit-exact against golden CRCs computed by the CPU oracle in the
build container (tests/golden/big_8320x40000_q95.json), plus size-independent properties (stock decoder, PSNR)."""
import io
import json
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_8320x40000_q95.json")))
W, H = 8320, 40000


@pytest.fixture(scope="module")
def big_image(mij):
    d = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    mij.synth_image_device(d.data_ptr(), W, 0, H, W * 3, bgr=False)
    torch.cuda.synchronize()
    # spot-check the device generator against the golden CRC of the whole synthetic image
    assert "%08x" % zlib.crc32(d.cpu().numpy().tobytes()) == GOLD["synthetic_crc32"]
    return d


@pytest.mark.parametrize("css,optimize,ri", [(1, True, -1), (0, True, -1), (2, True, -1), (3, True, -1), (4, True, -1), (1, False, -1),
                                             (1, True, 104), (2, True, 52), (3, True, 80)])
def test_fullsize_bit_exact(mij, big_image, css, optimize, ri):
    """ri -1: the interval the library picks (whole 64-block batches: 64 or 32 MCUs); the explicit ones are round 1's, which
    divide the MCU row instead."""
    with mij.Encoder(W, H, 95, optimize, css, restart_interval=ri) as enc:
        ri = enc.geometry["restart_interval"]
        enc.encode_device(big_image.data_ptr(), W * 3, "rgb")
        jpg = enc.retrieve()
    key = "css%d_ri%d_%s" % (css, ri, "opt" if optimize else "fix")
    assert key in GOLD["cases"], "no golden vector for %s (restart interval changed?)" % key
    assert len(jpg) == GOLD["cases"][key]["len"]
    assert "%08x" % zlib.crc32(jpg) == GOLD["cases"][key]["crc32"]


@pytest.mark.parametrize("ri", [-1, 104])
def test_fullsize_progressive_bit_exact(mij, big_image, ri):
    """Progressive output at the BASELINE size: the oracle's (= libjpeg's) file, by length and CRC -- with the interval the
    library picks for progressive output (a multiple of 64 blocks: 640 here) and with the baseline path's 104."""
    with mij.Encoder(W, H, 95, True, 1, restart_interval=ri, progressive=True) as enc:
        got_ri = enc.geometry["restart_interval"]
        assert got_ri == (640 if ri < 0 else ri)
        enc.encode_device(big_image.data_ptr(), W * 3, "rgb")
        jpg = enc.retrieve()
    gold = GOLD["cases"]["css1_ri%d_progressive" % got_ri]
    assert len(jpg) == gold["len"] and "%08x" % zlib.crc32(jpg) == gold["crc32"]


def test_fullsize_stock_decoder_psnr(mij, oracle, big_image):
    """Headline config (8320x40000 q95 4:2:2 optimised): decodes in libjpeg-turbo; PSNR equals libjpeg-turbo's own
    encode of the same input (31.162 dB) within the north-star's 0.05 dB."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with mij.Encoder(W, H, 95, True, 1) as enc:
        enc.encode_device(big_image.data_ptr(), W * 3, "rgb")
        jpg = enc.retrieve()
    dec = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))
    assert dec.shape == (H, W, 3)
    se = 0.0
    for y in range(0, H, 4000):
        src = big_image[y:y + 4000].cpu().numpy().astype(np.int32)
        se += float(((dec[y:y + 4000].astype(np.int32) - src) ** 2).sum())
    psnr = 10 * np.log10(255.0 ** 2 / (se / (3.0 * W * H)))
    assert abs(psnr - 31.162) < 0.05
    assert abs(len(jpg) / (3.0 * W * H) - 0.2040) < 0.001     # ratio; libjpeg-turbo without DRI: 0.2039


def test_every_rgb_colour(mij, oracle):
    """4096x4096 image holding each of the 2^24 RGB values once: the fp32 colour conversion of K1 must agree with the
    oracle's 16.16 fixed point for every input (checked through the quantised coefficients, q100, 4:4:4)."""
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)
    with mij.Encoder(4096, 4096, 100, False, 0) as enc:
        enc.encode_host(img, "rgb")
        got = enc.debug_coefficients()
    want = oracle.coefficients(img, 100, 0)
    assert np.array_equal(got, want)
    with mij.Encoder(4096, 4096, 100, False, 1) as enc:      # and through the h2v1 chroma path, BGR order
        enc.encode_host(np.ascontiguousarray(img[..., ::-1]), "bgr")
        got = enc.debug_coefficients()
    assert np.array_equal(got, oracle.coefficients(img, 100, 1))
