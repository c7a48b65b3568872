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


# ---------------------------------------------------------------------------------------------------------------------
# Progressive files WITHOUT restart markers through the parallel decoder (k_decode_prog.inc, round 4): first scans by subsequence
# synchronisation, AC refinement scans by hypothesis search over the history maps + exact verification, the wave decoder of
# k_decode_wave.inc as the on-device fallback. Whatever route a scan takes, the pixels are libjpeg-turbo's.
# ---------------------------------------------------------------------------------------------------------------------
def _grey_save(g, **kw):
    b = io.BytesIO()
    Image.fromarray(g, "L").save(b, "JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("ss", [0, 1, 2, "grey"])
@pytest.mark.parametrize("size,q", [((1234, 777), 95), ((2080, 1536), 96), ((3000, 520), 95), ((520, 3000), 98), ((1600, 1200), 88)])
def test_progressive_without_restart_markers_in_parallel(mij, oracle, ss, size, q):
    """Images large enough for several anchors per refinement scan (2,048 blocks apart) and hundreds of subsequences per first scan,
    all samplings Pillow can write and greyscale. On this dense content (noise + texture at q >= 95) the parallel decoder must take
    every scan it is tried on: a silent fall-back to the wave decoder would still be exact, and a hundred times slower. (At q88 the
    history maps of some refinement scans are too thin for every anchor to resolve: those scans may go to the wave decoder.)"""
    W, H = size
    img = oracle.synth_rgb(W, H)
    if ss == "grey":
        jpg = _grey_save(img[..., 1], quality=q, progressive=True, optimize=True)
    else:
        jpg = _save(img, quality=q, subsampling=ss, progressive=True, optimize=True)
    assert b"\xff\xc2" in jpg and b"\xff\xdd" not in jpg            # SOF2, no DRI
    with mij.Decoder() as dec:
        got = dec.decode_host(jpg, "rgb")
        tried, parallel = dec.px_report()
        assert np.array_equal(got, _pil_dec(jpg))
        assert tried >= 5 and (parallel == tried if q >= 95 else parallel >= 5), (tried, parallel)
        assert np.array_equal(dec.decode_host(jpg, "bgr"), _pil_dec(jpg)[..., ::-1])       # and again on the same handle


@pytest.mark.parametrize("size,q,ss", [((1234, 777), 75, 1), ((1040, 512), 75, 1), ((1040, 512), 90, 2), ((8320, 2048), 90, 2), ((8320, 2048), 75, 1)])
def test_progressive_thin_history_anchors_are_decided_by_the_chain(mij, oracle, size, q, ss):
    """Lower qualities: the history maps are thin, the paths that run a whole number of blocks beside the true one meet no violation
    and few anchors are unanimous. Their survivors stay on as candidates and the state the anchor before arrives at picks the true
    one (k_px_prewalk / k_px_chain); before that these files sent 3-4 of their 9 scans to the wave decoder. The 17-Mpixel files also
    need the sweeps (the finest level searching again next to what the chain has decided, with the scan's own fitted bits per block
    and per history bit) and what the search resolved demoted to candidates."""
    img = oracle.synth_rgb(*size)
    jpg = _save(img, quality=q, subsampling=ss, progressive=True, optimize=True)
    assert b"\xff\xc2" in jpg and b"\xff\xdd" not in jpg
    with mij.Decoder() as dec:
        got = dec.decode_host(jpg, "rgb")
        tried, parallel = dec.px_report()
        assert np.array_equal(got, _pil_dec(jpg))
        assert tried == 9 and parallel == tried, (tried, parallel)


def test_progressive_fallback_is_decided_per_scan_and_exact(mij, oracle):
    """Smooth content: the history maps of the refinement scans are nearly empty, almost every hypothesis parses without a violation
    and no anchor is unanimous -- those scans go to the wave decoder (on the device, no host decision), the first scans still run
    in parallel. Same pixels either way."""
    yy, xx = np.mgrid[0:900, 0:1400]
    smooth = np.stack([(xx * 255 // 1400), (yy * 255 // 900), ((xx + yy) * 255 // 2300)], -1).astype(np.uint8)
    jpg = _save(smooth, quality=90, subsampling=2, progressive=True, optimize=True)
    with mij.Decoder() as dec:
        got = dec.decode_host(jpg, "rgb")
        tried, parallel = dec.px_report()
        assert np.array_equal(got, _pil_dec(jpg))
        assert tried >= 9 and 5 <= parallel <= tried
    # spectral selection only (no successive approximation: no refinement scans at all): IJG's scan script is not reachable through
    # Pillow, but a quality-100 greyscale file has Al = 0 first scans of every band plus refinements; covered above. Flat image:
    flat = np.full((800, 1200, 3), 128, np.uint8)
    flat[200:600, 300:900] = (250, 20, 20)
    jpg = _save(flat, quality=85, subsampling=1, progressive=True)
    with mij.Decoder() as dec:
        assert np.array_equal(dec.decode_host(jpg, "rgb"), _pil_dec(jpg))          # long end-of-band runs across subsequences


def test_progressive_parallel_switches():
    """MIJ_PROG_PARALLEL=0 keeps rounds 2-3's wave decoder for every scan, MIJ_PX_LANE_REFINE=1 the first (lane per segment) form of
    the verify pass: all three give libjpeg-turbo's pixels. The switches are read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import io, sys, numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 26
sys.path.insert(0, %r)
import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O
img = O.synth_rgb(1234, 777)
with mij.Decoder() as dec:
    for kw in (dict(quality=95, subsampling=1), dict(quality=97, subsampling=2)):
        b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", progressive=True, optimize=True, **kw); j = b.getvalue()
        assert np.array_equal(dec.decode_host(j, "rgb"), np.asarray(Image.open(io.BytesIO(j)).convert("RGB")))
        print("report", dec.px_report())
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # MIJ_PX_WS_BUDGET_MB (round 5): the parallel decoder's workspace above the budget -> those scans are walked by the wave decoder
    # instead of the decode failing (0: none of it may be used; 1: this file's ~9 MB per refinement scan does not fit, the first scans' do)
    for env, want in (({"MIJ_PROG_PARALLEL": "0"}, "report (0, 0)"), ({"MIJ_PX_LANE_REFINE": "1"}, "report (9, 9)"), ({}, "report (9, 9)"),
                      ({"MIJ_PX_WS_BUDGET_MB": "0"}, "report (0, 0)"), ({"MIJ_PX_WS_BUDGET_MB": "1"}, "report (")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout and want in r.stdout, (env, r.stdout[-500:], r.stderr[-2000:])


def test_px_report_speaks_of_the_last_decode(mij, oracle):
    """A progressive decode followed by a baseline decode on the same handle: the report is the baseline file's (no progressive scans)."""
    img = oracle.synth_rgb(640, 480)
    with mij.Decoder() as dec:
        jp = _save(img, quality=95, subsampling=1, progressive=True, optimize=True)
        assert np.array_equal(dec.decode_host(jp, "rgb"), _pil_dec(jp))
        assert dec.px_report()[0] > 0
        jb = _save(img, quality=95, subsampling=1)
        assert np.array_equal(dec.decode_host(jb, "rgb"), _pil_dec(jb))
        assert dec.px_report() == (0, 0)


def test_corrupt_progressive_file_same_answer_on_both_routes():
    """A progressive file without restart markers whose first scans carry invalid codes: the parallel decoder's exact write pass sends the
    scan to the wave decoder, which reports it -- the return code does not depend on which route a scan took (MIJ_PROG_PARALLEL=0 / 1)."""
    import os
    import subprocess
    import sys
    code = r'''
import io, sys, numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 26
sys.path.insert(0, %r)
import nvjpeg_imagecompressor_amd as mij
from oracle import oracle as O
img = O.synth_rgb(800, 600)
b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", progressive=True, optimize=False, quality=90, subsampling=1); j = bytearray(b.getvalue())
# find the first SOS (DC scan) and the luma AC first scan (second SOS with one component); overwrite a stretch in the middle of each
# with 0xFE bytes: sixteen 1-bits in a row is no code of a standard table (and no marker)
out = []
pos = [i for i in range(len(j) - 1) if j[i] == 0xFF and j[i + 1] == 0xDA]
for which in (0, 1):
    k = bytearray(j)
    a, e = pos[which], pos[which + 1]
    mid = (a + e) // 2
    k[mid:mid + 64] = b"\xfe" * 64
    with mij.Decoder() as dec:
        try:
            dec.decode_host(bytes(k), "rgb")
            out.append("ok")
        except mij.MiJpegError as ex:
            out.append("err")
print("result", out)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MIJ_PROG_PARALLEL=v), timeout=600)
        assert r.returncode == 0 and "result" in r.stdout, (v, r.stdout[-500:], r.stderr[-2000:])
        got[v] = r.stdout[r.stdout.index("result"):].strip()
    assert got["0"] == got["1"], got
    assert "err" in got["0"], got           # sixteen 1-bits is an invalid code: at least one of the two files must be refused


def test_fullsize_progressive_file_without_restart_markers(mij):
    """Round 3's criterion: a libjpeg-turbo PROGRESSIVE 8320x40000 q95 4:2:2 file with NO DRI -- the reference's own output format
    (ImageCompressorImpl.cu:28) at its headline size -- decodes pixel-identically, device resident, in a fraction of a second
    (rounds 2-3: ~80 s, one wave per scan). Pillow writes the file here (~10 s); its CRC and the CRC of the decoded image are the
    committed golden (tests/golden/prog_nodri_8320x40000.json, from tools/decode_prog_nodri_fullsize.py)."""
    import json
    import os
    import zlib
    import torch
    ImageFile.MAXBLOCK = 1 << 30
    Image.MAX_IMAGE_PIXELS = None
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prog_nodri_8320x40000.json")))
    W, H = 8320, 40000
    d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    mij.synth_image_device(d_img.data_ptr(), W, 0, H, W * 3, bgr=False)
    torch.cuda.synchronize()
    b = io.BytesIO()
    Image.fromarray(d_img.cpu().numpy()).save(b, "JPEG", quality=95, subsampling=1, progressive=True, optimize=True)
    jpg = b.getvalue()
    assert (len(jpg), "%08x" % zlib.crc32(jpg)) == (gold["file_bytes"], gold["file_crc32"])
    d_file = torch.frombuffer(bytearray(jpg), dtype=torch.uint8).cuda()
    d_out = torch.empty_like(d_img)
    with mij.Decoder() as dec:
        ms = []
        for _ in range(3):
            dec.decode_device_ptr(d_file.data_ptr(), len(jpg), d_out.data_ptr(), W * 3, "rgb")
            ms.append(dec.sync())
        tried, parallel = dec.px_report()
    assert "%08x" % zlib.crc32(d_out.cpu().numpy().tobytes()) == gold["decoded_crc32"]      # = Pillow's own decode of the file
    assert (tried, parallel) == (9, 9)
    assert min(ms) < 1000.0, ms                 # (measured ~0.2 s; the bound only says "not the serial route": that one takes ~80 s)
    print("full-size progressive no-DRI decode: %s ms" % ["%.1f" % m for m in ms])
