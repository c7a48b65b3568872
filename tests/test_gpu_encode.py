"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle (bit exact)."""
import io
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CSS_ALL = [0, 1, 2, 3, 4, 5]


def test_the_library_on_this_box_is_the_tree(mij):
    """On the GPU box: the libmijpeg.so that travelled with the snapshot was built from exactly these sources (checked when
    pytest started, before any fixture could rebuild it), and the library this process has loaded says so itself."""
    from conftest import SHIPPED
    from nvjpeg_imagecompressor_amd import build as B
    assert "error" not in SHIPPED, SHIPPED
    assert SHIPPED["present"] and not SHIPPED["stale"], "the shipped libmijpeg.so did not match the sources: %s" % SHIPPED
    assert mij.library_source_hash() == B.source_hash() == SHIPPED["tree_hash"]


def _img(oracle, W, H, kind, seed=0):
    if kind == "synth":
        return oracle.synth_rgb(W, H)
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == "flat":
        return np.full((H, W, 3), 77, np.uint8)
    raise ValueError(kind)


@pytest.mark.parametrize("css", CSS_ALL)
@pytest.mark.parametrize("size", [(512, 512), (64, 48), (8, 8), (1, 1), (17, 33), (100, 75), (129, 65), (250, 3), (1040, 136)])
def test_coefficients_match_oracle(mij, oracle, css, size):
    W, H = size
    img = _img(oracle, W, H, "synth")
    with mij.Encoder(W, H, 95, True, css) as enc:
        enc.encode_host(img, "rgb")
        got = enc.debug_coefficients()
    want = oracle.coefficients(img, 95, css)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (mcu, blk, k): %s" % bad[:5].tolist()


@pytest.mark.parametrize("css", CSS_ALL)
@pytest.mark.parametrize("optimize", [False, True])
@pytest.mark.parametrize("size,kind", [((512, 512), "synth"), ((100, 75), "noise"), ((17, 33), "synth"), ((64, 64), "flat"),
                                       ((1040, 136), "noise")])
def test_file_bytes_match_oracle(mij, oracle, css, optimize, size, kind):
    W, H = size
    img = _img(oracle, W, H, kind, seed=css)
    for q in (95, 30, 100):
        with mij.Encoder(W, H, q, optimize, css) as enc:
            ri = enc.geometry["restart_interval"]
            got = enc.encode_host(img, "rgb")
        want = oracle.encode(img, q, css, optimize, ri)
        assert len(got) == len(want), (q, len(got), len(want))
        assert got == want, "q=%d first diff at %d" % (q, next(i for i in range(len(want)) if got[i] != want[i]))


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgb_planar", "bgr_planar"])
def test_input_formats(mij, oracle, fmt):
    W, H = 136, 72
    rgb = _img(oracle, W, H, "synth")
    if fmt == "rgb":
        arr = rgb
    elif fmt == "bgr":
        arr = rgb[..., ::-1]
    elif fmt == "rgb_planar":
        arr = rgb.transpose(2, 0, 1)
    else:
        arr = rgb[..., ::-1].transpose(2, 0, 1)
    with mij.Encoder(W, H, 95, True, 1) as enc:
        ri = enc.geometry["restart_interval"]
        got = enc.encode_host(np.ascontiguousarray(arr), fmt)
    assert got == oracle.encode(rgb, 95, 1, True, ri)


def test_explicit_restart_intervals(mij, oracle):
    W, H = 256, 64
    img = _img(oracle, W, H, "noise", 3)
    for ri in (1, 3, 7, 16, 64, 1000):
        with mij.Encoder(W, H, 90, True, 2, restart_interval=ri) as enc:
            got = enc.encode_host(img, "rgb")
        assert got == oracle.encode(img, 90, 2, True, ri), ri


def test_stock_decoder_roundtrip_and_psnr(mij, oracle):
    from PIL import Image
    W, H = 512, 512
    img = _img(oracle, W, H, "synth")
    with mij.Encoder(W, H, 95, False, 0, restart_interval=64) as enc:
        got = enc.encode_host(img, "rgb")
    dec = np.asarray(Image.open(io.BytesIO(got)).convert("RGB"))
    # config 1 of BASELINE.json: libjpeg-turbo gives 35.688 dB on this input; same coefficients => same PSNR
    assert abs(oracle.psnr(img, dec) - 35.688) < 0.05
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=95, subsampling=0, restart_marker_blocks=64)
    assert got == b.getvalue()


def test_repeat_encodes_are_identical_and_timed(mij, oracle):
    W, H = 512, 256
    img = _img(oracle, W, H, "synth")
    with mij.Encoder(W, H, 95, True, 1) as enc:
        enc.enable_timing(True)
        a = enc.encode_host(img, "rgb")
        b = enc.encode_host(img, "rgb")
        t = enc.stage_times()
    assert a == b
    assert t["total"] > 0


def test_reference_facade(mij, oracle, tmp_path, capsys):
    W, H = 208, 120
    bgr = _img(oracle, W, H, "synth")[..., ::-1]
    r = mij.NvjpegCompressRunner(W, H, 95, True)
    r.buildCompressEnv()
    out, state = r.compress(np.ascontiguousarray(bgr))
    assert state == 1 and out[:2] == b"\xff\xd8" and out[-2:] == b"\xff\xd9"
    bad, state = r.compress(np.zeros((H + 1, W, 3), np.uint8))  # wrong size: reference overruns, we refuse
    assert state == 0 and bad == b""
    r.save(str(tmp_path / "o.jpg"), out)
    assert (tmp_path / "o.jpg").read_bytes() == out
    r.deleteCompressEnv()
    assert "Compress Cost time" in capsys.readouterr().out


def _golden_cases():
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    idx = json.load(open(os.path.join(gold, "index.json")))
    return gold, [c for c in idx["cases"] if c["file"] and c["restart"] > 0 and c["encoder"] == "libjpeg-turbo"]


@pytest.mark.parametrize("case", _golden_cases()[1], ids=lambda c: c["file"])
def test_gpu_output_equals_committed_libjpeg_turbo_files(mij, oracle, case):
    """The HIP path against files written by libjpeg-turbo itself (committed under tests/golden/), whole file, baseline and
    progressive -- no oracle in between. (Files without DRI cannot be asked of this encoder: the restart interval is its
    unit of parallelism.)"""
    import os
    w, h = map(int, case["size"].split("x"))
    img = oracle.synth_rgb(w, h)
    want = open(os.path.join(_golden_cases()[0], case["file"]), "rb").read()
    with mij.Encoder(w, h, case["quality"], case["optimize"], case["css"], restart_interval=case["restart"],
                     progressive=bool(case.get("progressive"))) as enc:
        assert enc.encode_host(img, "rgb") == want


def test_fast_coder_falls_back_from_narrow_strips(mij, oracle):
    """The fast entropy coder starts on 16-word strips (512 bits per block); noise at q95 overflows them in every restart
    interval, the roomy coder takes those intervals over, and the handle moves to 24-word strips for the next image. A
    picture in between (few or no overflows) must come out right on either. Every file equals the oracle's."""
    rng = np.random.default_rng(21)
    W, H = 640, 384
    noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    half = noise.copy()
    half[:, : W // 2] = oracle.synth_rgb(W, H)[:, : W // 2]
    calm = oracle.synth_rgb(W, H)
    for css in (0, 1, 2):
        with mij.Encoder(W, H, 95, True, css) as enc:
            ri = enc.geometry["restart_interval"]
            for img in (calm, noise, half, calm, noise):
                assert enc.encode_host(img, "rgb") == oracle.encode(img, 95, css, True, ri)


def test_output_larger_than_the_preallocated_buffer(mij, oracle):
    """Black/white noise at q100 with a restart marker after every MCU: more than one byte per coefficient, the initial
    capacity. The encoder grows its output buffer and redoes header + compaction (mij_encode_result)."""
    rng = np.random.default_rng(9)
    W, H = 1280, 400
    img = rng.integers(0, 2, (H, W, 3), dtype=np.uint8) * 255
    for optimize in (True, False):
        with mij.Encoder(W, H, 100, optimize, 0, restart_interval=1) as enc:
            got = enc.encode_host(img, "rgb")
            assert len(got) > (W // 8) * (H // 8) * 3 * 64 + 65536
            assert got == oracle.encode(img, 100, 0, optimize, 1)
            assert enc.encode_host(img, "rgb") == got          # and again, now that the buffer is large enough


def test_fused_entropy_coder_opt_in_matches_oracle(oracle, tmp_path):
    """MIJ_FUSE=1 (opt-in experiment): K4 with the size scan and the stuffing + compaction folded in by a decoupled
    look-back. Slower than the three kernels, but it must stay exact, including an image whose blocks overflow the fast
    coder's strips (noise at q100) and a buffer that has to grow."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, zlib, numpy as np
sys.path.insert(0, %r)
import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O
rng = np.random.default_rng(5)
for (W, H, q, css, img) in [(2080, 1000, 95, 1, O.synth_rgb(2080, 1000)), (416, 248, 100, 0, rng.integers(0, 256, (248, 416, 3), dtype=np.uint8)),
                            (333, 77, 35, 2, O.synth_rgb(333, 77))]:
    with mij.Encoder(W, H, q, True, css) as enc:
        got = enc.encode_host(img, "rgb")
        ri = enc.geometry["restart_interval"]
    want = O.encode(img, q, css, True, ri)
    assert got == want, (W, H, q, css, len(got), len(want))
print("fused ok")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MIJ_FUSE="1"), timeout=300)
    assert r.returncode == 0 and "fused ok" in r.stdout, r.stdout + r.stderr


def test_tables_on_a_second_stream(mij, oracle):
    """mij_encode_tables: the table build of image A issued on a side stream between A's transform and B's entropy coder (the
    bench's loop). Files must be the oracle's, with alternating images so that a table built for the wrong image would show."""
    import torch
    W, H = 1040, 520
    imgs = [oracle.synth_rgb(W, H), np.random.default_rng(4).integers(0, 256, (H, W, 3), dtype=np.uint8)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    main, side = torch.cuda.current_stream().cuda_stream, torch.cuda.Stream()
    encs = [mij.Encoder(W, H, 90, True, 1) for _ in range(3)]
    for e in encs:
        e.enable_timing(True)
    ri = encs[0].geometry["restart_interval"]
    want = [oracle.encode(i, 90, 1, True, ri) for i in imgs]
    got, waiting, coded = [], None, None
    for n in range(7):
        e = encs[n % 3]
        e.transform(d[n & 1].data_ptr(), W * 3, "rgb", 0, main)
        e.tables(side.cuda_stream)
        if waiting is not None:
            waiting.entropy(main)
        if coded is not None:
            got.append(coded.retrieve())
            assert coded.stage_times()["entropy"] > 0
        coded, waiting = waiting, e
    got.append(coded.retrieve())
    waiting.entropy(main)
    got.append(waiting.retrieve())
    assert got == [want[n & 1] for n in range(7)]
    for e in encs:
        e.close()
