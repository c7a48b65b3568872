"""Progressive (SOF2) output -- the encoding the reference configures (ImageCompressorImpl.cu:28) -- through the C ABI,
byte-for-byte against the oracle's restatement of libjpeg's progressive encoder (itself byte-identical to libjpeg-turbo,
tests/test_oracle_pin.py), plus directly against Pillow where Pillow can be given the same restart interval."""
import io

import numpy as np
import pytest
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 26
pytestmark = pytest.mark.gpu


def _images(oracle, W, H, seed):
    rng = np.random.default_rng(seed)
    yield oracle.synth_rgb(W, H)
    yield rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    flat = np.full((H, W, 3), 200, np.uint8)
    flat[H // 3:, W // 2:] = (10, 90, 250)
    yield flat


@pytest.mark.parametrize("css", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("size", [(64, 64), (17, 33), (200, 136), (1, 1), (333, 77), (1040, 136)])
def test_progressive_matches_oracle(mij, oracle, css, size):
    W, H = size
    for n, img in enumerate(_images(oracle, W, H, css)):
        for q in (95, 35, 100):
            with mij.Encoder(W, H, q, True, css, progressive=True) as enc:
                ri = enc.geometry["restart_interval"]
                got = enc.encode_host(img, "rgb")
            want = oracle.encode_progressive(img, q, css, ri)
            assert len(got) == len(want), (n, q, len(got), len(want))
            assert got == want, (n, q, next(i for i in range(len(want)) if got[i] != want[i]))


@pytest.mark.parametrize("ri", [1, 7, 40])
def test_progressive_matches_libjpeg_turbo_directly(mij, oracle, ri):
    img = oracle.synth_rgb(176, 120)
    for css in (0, 1, 2):
        with mij.Encoder(176, 120, 85, True, css, restart_interval=ri, progressive=True) as enc:
            got = enc.encode_host(img, "rgb")
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=85, subsampling=css, progressive=True, restart_marker_blocks=ri)
        assert got == b.getvalue()
        dec = np.asarray(Image.open(io.BytesIO(got)).convert("RGB"))
        assert dec.shape == img.shape


def test_progressive_smaller_than_baseline_and_same_pixels(mij, oracle):
    W, H = 2080, 1000
    img = oracle.synth_rgb(W, H)
    with mij.Encoder(W, H, 95, True, 1) as enc:
        base = enc.encode_host(img, "rgb")
    with mij.Encoder(W, H, 95, True, 1, progressive=True) as enc:
        prog = enc.encode_host(img, "rgb")
        assert prog == oracle.encode_progressive(img, 95, 1, enc.geometry["restart_interval"])
    assert len(prog) < len(base)
    a = np.asarray(Image.open(io.BytesIO(base)).convert("RGB"))
    b = np.asarray(Image.open(io.BytesIO(prog)).convert("RGB"))
    assert np.array_equal(a, b)          # same quantised coefficients
    with mij.Decoder() as dec:           # and our own decoder reads it back
        assert np.array_equal(dec.decode_host(prog, "rgb"), b)


def test_progressive_strips_need_an_aligned_interval(mij):
    """Round 5: a progressive encoder may own a strip when every scan's restart intervals end at the strip's boundaries -- the interval
    divides the MCUs per row and the width is a whole number of MCUs (tests/test_gpu_sharded.py encodes with such strips)."""
    with pytest.raises(mij.MiJpegError, match="divides the MCUs per row"):
        mij.Encoder(2080, 1000, 95, True, 1, restart_interval=64, strip_mcu_row0=0, strip_mcu_rows=32, progressive=True)    # 130 MCUs per row: 32 rows are whole intervals of the interleaved scans, not of the luma scans
    with pytest.raises(mij.MiJpegError, match="whole MCUs"):
        mij.Encoder(520, 512, 90, True, 1, restart_interval=33, strip_mcu_row0=0, strip_mcu_rows=8, progressive=True)       # 65 luma blocks, 33 MCUs per row
    with mij.Encoder(512, 512, 90, True, 0, strip_mcu_row0=8, strip_mcu_rows=8, progressive=True) as e:                    # AUTO = 64 = the MCU row here
        assert e.geometry["restart_interval"] == 64 and e.geometry["strip_y0"] == 64


def test_facade_progressive(mij, oracle, capsys):
    """NvjpegCompressRunner mirror with the reference's encoding (ImageCompressorImpl.cu:28) switched on."""
    W, H = 208, 120
    bgr = np.ascontiguousarray(oracle.synth_rgb(W, H)[..., ::-1])
    r = mij.NvjpegCompressRunner(W, H, 95, True, css=1, verbose=True, progressive=True)
    r.buildCompressEnv()
    out, state = r.compress(bgr)
    assert state == 1 and b"\xff\xc2" in out[:700]
    assert "Compress Cost time" in capsys.readouterr().out      # the reference's timing line (ImageCompressorImpl.cu:289-291)
    r.deleteCompressEnv()
    b = io.BytesIO()
    Image.fromarray(bgr[..., ::-1]).save(b, "JPEG", quality=95, subsampling=1, progressive=True,
                                         restart_marker_blocks=mij.geometry_query(W, H, 95, True, 1, progressive=True)["restart_interval"])
    assert out == b.getvalue()


def test_progressive_eob_run_cap_and_correction_flush(mij, oracle):
    """Restart intervals long enough for an EOB run to reach 0x7FFF blocks, and noise at q100 for the 937-bit rule."""
    W, H = 2304, 2048
    img = np.full((H, W, 3), (120, 200, 33), np.uint8)
    img[1000:1010, 500:520] = 255
    with mij.Encoder(W, H, 90, True, 0, restart_interval=65535, progressive=True) as enc:
        got = enc.encode_host(img, "rgb")
    assert got == oracle.encode_progressive(img, 90, 0, 65535)
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=0, progressive=True, restart_marker_blocks=65535)
    assert got == b.getvalue()
    noise = np.random.default_rng(5).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    for q in (100, 98):
        with mij.Encoder(512, 512, q, True, 0, progressive=True) as enc:
            assert enc.encode_host(noise, "rgb") == oracle.encode_progressive(noise, q, 0, enc.geometry["restart_interval"])


def test_progressive_output_larger_than_the_preallocated_buffer(mij, oracle):
    """Noise at q100 with a restart marker after every block of every scan: the file outgrows 2 bytes per coefficient."""
    rng = np.random.default_rng(9)
    W, H = 1280, 400
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    with mij.Encoder(W, H, 100, True, 0, restart_interval=1, progressive=True) as enc:
        got = enc.encode_host(img, "rgb")
    assert got == oracle.encode_progressive(img, 100, 0, 1)
    assert len(got) > (W // 8) * (H // 8) * 3 * 64 + 65536      # really beyond the initial capacity (2 B per coefficient)
    dec = np.asarray(Image.open(io.BytesIO(got)).convert("RGB"))
    assert dec.shape == img.shape


@pytest.mark.parametrize("q", [95, 85, 60])
def test_refinement_scans_with_narrow_strips_and_after_the_switch_to_wide(mij, oracle, q):
    """The emit pass of a refinement scan starts on 16-word strips; blocks that overflow them go to the serial kernel, and
    a scan that overflows often is coded with 24-word strips from the next image on: both files must be the oracle's."""
    W, H = 1536, 1024
    rng = np.random.default_rng(q)
    noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    busy = oracle.synth_rgb(W, H)
    busy[::2, 1::2] ^= 0x3F                                        # fine texture on the synthetic image
    for css in (1, 0):
        with mij.Encoder(W, H, q, True, css, progressive=True) as enc:
            ri = enc.geometry["restart_interval"]
            for img in (noise, noise, busy, noise):
                assert enc.encode_host(img, "rgb") == oracle.encode_progressive(img, q, css, ri)
