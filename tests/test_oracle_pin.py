"""Pins the CPU oracle (oracle/jpeg_oracle.c) against the committed golden vectors and, live, against the stock
encoders in this image: libjpeg-turbo (Pillow) whole-file byte equality, IJG libjpeg 9d scan-data equality.
The reference has no fixtures of its own: this is the substitute pin."""
import io
import json
import os
import subprocess
import zlib

import numpy as np
import pytest
from PIL import Image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))


def _scan(j):
    i = j.index(b"\xff\xda")
    return j[i + 2 + ((j[i + 2] << 8) | j[i + 3]):]


def test_synthetic_generator_pins(oracle):
    img = oracle.synth_rgb(512, 512)
    assert "%08x" % zlib.crc32(img.tobytes()) == "6f0fd7dc"          # SURVEY.md 8(d) pin
    assert img[0, :4].tolist() == [[32, 35, 53], [46, 42, 79], [44, 30, 50], [52, 43, 48]]
    assert np.array_equal(img, oracle.synth_rgb_numpy(512, 512))
    strip = oracle.synth_rgb(512, 512, y0=100, rows=17)
    assert np.array_equal(strip, img[100:117])
    for k, v in INDEX["synthetic_crc32"].items():
        w, h = map(int, k.split("x"))
        assert "%08x" % zlib.crc32(oracle.synth_rgb(w, h).tobytes()) == v


@pytest.mark.parametrize("case", [c for c in INDEX["cases"] if c["file"]], ids=lambda c: c["file"])
def test_golden_files(oracle, case):
    w, h = map(int, case["size"].split("x"))
    img = oracle.synth_rgb(w, h)
    want = open(os.path.join(GOLD, case["file"]), "rb").read()
    assert len(want) == case["len"] and "%08x" % zlib.crc32(want) == case["crc32"]
    if case.get("progressive"):
        got = oracle.encode_progressive(img, case["quality"], case["css"], case["restart"])
    else:
        got = oracle.encode(img, case["quality"], case["css"], case["optimize"], case["restart"])
    if case["encoder"] == "libjpeg-turbo":
        assert got == want                     # whole file, headers included
    else:
        assert _scan(got) == _scan(want)       # IJG writes its markers in a different order; the coded data is equal


@pytest.mark.parametrize("case", [c for c in INDEX["cases"] if not c["file"]],
                         ids=lambda c: "512_css%d_%s_rst%d" % (c["css"], "opt" if c["optimize"] else "fix", c["restart"]))
def test_golden_512(oracle, case):
    img = oracle.synth_rgb(512, 512)
    got = oracle.encode(img, 95, case["css"], case["optimize"], case["restart"])
    assert len(got) == case["len"] and "%08x" % zlib.crc32(got) == case["crc32"]
    dec = np.asarray(Image.open(io.BytesIO(got)).convert("RGB"))
    assert abs(oracle.psnr(img, dec) - case["psnr"]) < 1e-3


def test_baseline_config1(oracle):
    """BASELINE.json configs[0]: 512x512 q95 4:4:4 fixed Huffman -> 265,098 B, CRC 03d11f00, 35.688 dB."""
    img = oracle.synth_rgb(512, 512)
    j = oracle.encode(img, 95, 0, False, 0)
    assert len(j) == 265098 and "%08x" % zlib.crc32(j) == "03d11f00"
    dec = np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))
    assert abs(oracle.psnr(img, dec) - 35.688) < 0.001


def _pil(img, q, ss, opt, rst):
    b = io.BytesIO()
    kw = dict(quality=q, subsampling=ss, optimize=opt)
    if rst:
        kw["restart_marker_blocks"] = rst
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (9, 9), (16, 16), (17, 33), (31, 8), (100, 75), (129, 65), (250, 3)])
def test_live_vs_libjpeg_turbo(oracle, size):
    """Edge padding, dummy blocks, all qualities, DRI, optimal tables: byte-identical files."""
    W, H = size
    rng = np.random.default_rng(W * 1000 + H)
    imgs = [oracle.synth_rgb(W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8), np.full((H, W, 3), 200, np.uint8)]
    for img in imgs:
        for css in (0, 1, 2):
            for q in (1, 50, 95, 100):
                for opt in (False, True):
                    for rst in (0, 3):
                        assert oracle.encode(img, q, css, opt, rst) == _pil(img, q, css, opt, rst), (css, q, opt, rst)


def test_live_vs_ijg_all_samplings(oracle, tmp_path):
    """4:4:0 / 4:1:1 / 4:1:0 (and the others again) against an independent code base (IJG 9d, YCbCr input)."""
    harness = os.path.join(os.path.dirname(oracle.__file__), "ijg_harness")
    if not os.path.exists(harness):
        pytest.skip("IJG libjpeg headers not present in this image")
    rng = np.random.default_rng(7)
    for (W, H) in [(17, 33), (100, 75), (64, 48)]:
        for img in (oracle.synth_rgb(W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8)):
            raw = tmp_path / "in.raw"
            oracle.rgb_to_ycc(img).tofile(str(raw))
            for css in range(6):
                hs, vs = oracle.CSS_FACTORS[css]
                for q in (20, 95):
                    for rst in (0, 3):
                        out = tmp_path / "o.jpg"
                        subprocess.check_call([harness, "enc", str(raw), str(W), str(H), "ycc", str(q), str(hs), str(vs),
                                               "0", str(rst), str(out)])
                        assert _scan(out.read_bytes()) == _scan(oracle.encode(img, q, css, False, rst)), (css, q, rst)
                    # optimised tables: IJG 9d builds different (also valid) tables; the coefficients it reads back
                    # from OUR file must equal the ones in ITS file
                    ours = tmp_path / "ours.jpg"
                    ours.write_bytes(oracle.encode(img, q, css, True, 0))
                    theirs = tmp_path / "theirs.jpg"
                    subprocess.check_call([harness, "enc", str(raw), str(W), str(H), "ycc", str(q), str(hs), str(vs), "1",
                                           "0", str(theirs)])
                    subprocess.check_call([harness, "coef", str(ours), str(tmp_path / "a.bin")])
                    subprocess.check_call([harness, "coef", str(theirs), str(tmp_path / "b.bin")])
                    assert (tmp_path / "a.bin").read_bytes() == (tmp_path / "b.bin").read_bytes()


def test_stage_apis_are_consistent(oracle):
    """coefficients -> histogram -> tables -> entropy coding compose to the same file as the one-shot encode."""
    img = oracle.synth_rgb(100, 75)
    for css in range(6):
        coef = oracle.coefficients(img, 90, css)
        g = oracle.geometry(100, 75, css)
        assert coef.shape == (g["mcux"] * g["mcuy"], g["bpm"], 64)
        for opt in (False, True):
            assert oracle.encode_coefficients(coef, 100, 75, 90, css, opt, 5) == oracle.encode(img, 90, css, opt, 5)
        hist = oracle.histogram(coef, 100, 75, css, 5)
        assert hist[0].sum() == g["mcux"] * g["mcuy"] * g["hs"] * g["vs"]      # one DC symbol per luma block
        assert hist[2].sum() == g["mcux"] * g["mcuy"] * 2
        bits, vals = oracle.gen_optimal_table(hist[1])
        assert bits[1:17].sum() == len(vals) == np.count_nonzero(hist[1][:256])
        # Kraft inequality with the reserved all-ones code point
        assert sum(int(bits[l]) * 2 ** (16 - l) for l in range(1, 17)) <= 2 ** 16 - 1
    # planar and BGR inputs describe the same picture
    a = oracle.encode(img, 95, 1, True, 0, "rgb")
    assert a == oracle.encode(img[..., ::-1], 95, 1, True, 0, "bgr")
    assert a == oracle.encode(img.transpose(2, 0, 1), 95, 1, True, 0, "rgb_planar")
    assert a == oracle.encode(img[..., ::-1].transpose(2, 0, 1), 95, 1, True, 0, "bgr_planar")


# ---- progressive mode (reference ImageCompressorImpl.cu:28): oracle restatement of libjpeg's scan script + jcphuff procedures
def _pil_progressive(img, q, ss, **kw):
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 26)     # libjpeg cannot suspend while writing progressive output
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=ss, progressive=True, **kw)
    return b.getvalue()


@pytest.mark.parametrize("size", [(64, 64), (17, 33), (200, 136), (1, 1), (333, 77)])
@pytest.mark.parametrize("ss", [0, 1, 2])
def test_progressive_bytes_equal_libjpeg_turbo(oracle, size, ss):
    """Byte-identical files: scan script, EOB runs, correction-bit buffering and flush rules, one optimal table per scan."""
    W, H = size
    rng = np.random.default_rng(W + ss)
    flat = np.full((H, W, 3), 200, np.uint8)
    flat[H // 3:, W // 2:] = (10, 90, 250)
    for img in (oracle.synth_rgb(W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8), flat):
        for q in (95, 35, 100):
            assert oracle.encode_progressive(img, q, ss) == _pil_progressive(img, q, ss), (q,)


@pytest.mark.parametrize("ri", [1, 7, 40])
def test_progressive_with_restart_interval(oracle, ri):
    img = oracle.synth_rgb(176, 120)
    for ss in (0, 1, 2):
        assert oracle.encode_progressive(img, 85, ss, ri) == _pil_progressive(img, 85, ss, restart_marker_blocks=ri)


def test_progressive_flush_rules(oracle):
    """The two flush rules of the EOB-run machinery: a run is cut at 0x7FFF blocks (large flat image, no DRI) and when more
    than 937 correction bits are waiting behind it (noise at q98-100)."""
    img = np.full((2048, 2304, 3), (120, 200, 33), np.uint8)
    img[1000:1010, 500:520] = 255
    for ss in (0, 2):
        assert oracle.encode_progressive(img, 90, ss) == _pil_progressive(img, 90, ss)
    noise = np.random.default_rng(5).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    for q in (100, 98):
        assert oracle.encode_progressive(noise, q, 0) == _pil_progressive(noise, q, 0)
