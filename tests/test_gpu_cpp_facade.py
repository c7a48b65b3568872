"""The C++ drop-in class (include/ImageCompressor.h, same surface as the reference's
src/ImageCompressorDll/ImageCompressor.h:22-42) driven by the demo that follows the reference main.cpp sequence."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "nvjpeg_imagecompressor_amd", "cpp")


def _dri(jpg):
    i = jpg.index(b"\xff\xdd")
    return (jpg[i + 4] << 8) | jpg[i + 5]


def test_demo_sequence_matches_oracle(mij, oracle, tmp_path):
    subprocess.check_call(["make", "-s", "-C", CPP, "all"])
    W, H = 416, 248
    imgs = [oracle.synth_rgb(W, H), oracle.synth_rgb(W, H + 100)[100:]]
    paths = []
    for i, im in enumerate(imgs):
        p = tmp_path / ("in%d.ppm" % i)
        with open(p, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (W, H))
            f.write(np.ascontiguousarray(im).tobytes())
        paths.append(str(p))
    for css in (0, 1):
        out = str(tmp_path / ("o%d" % css))
        r = subprocess.run([os.path.join(CPP, "demo"), paths[0], paths[1], "--css", str(css), "--out", out],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("[INFO] Successful.") == 4      # 2 compress + 2 decode
        assert "=> Compress Cost time" in r.stdout and "NvjpegCompressRunner Compress Func Cost Time" in r.stdout
        assert r.stdout.count("=> Decode Cost time : ") == 2 and "NvjpegCompressRunner Decode Func Cost Time" in r.stdout   # reference .cu:373, .cpp:85
        assert "Delete NvjpegCompressRunnerImpl Successfully" in r.stdout
        for i, im in enumerate(imgs):
            got = open("%s_%d.jpeg" % (out, i + 1), "rb").read()
            assert got == oracle.encode(im, 95, css, True, _dri(got))
            ppm = open("%s_%d_decode.ppm" % (out, i + 1), "rb").read()
            hdr = b"P6\n%d %d\n255\n" % (W, H)
            assert ppm.startswith(hdr)
            dec = np.frombuffer(ppm[len(hdr):], np.uint8).reshape(H, W, 3)
            assert np.array_equal(dec, oracle.decode(got))


def test_wrong_size_is_refused_not_overrun(mij, tmp_path):
    """The reference copies whatever Mat arrives into planes sized by the constructor (ImageCompressorImpl.cu:275,280);
    the drop-in returns run_state 0 and an empty vector instead."""
    subprocess.check_call(["make", "-s", "-C", CPP, "all"])
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    a.write_bytes(b"P6\n64 64\n255\n" + bytes(64 * 64 * 3))
    b.write_bytes(b"P6\n64 32\n255\n" + bytes(64 * 32 * 3))
    r = subprocess.run([os.path.join(CPP, "demo"), str(a), str(b), "--out", str(tmp_path / "x")], capture_output=True, text=True)
    assert r.returncode != 0
