"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950 without a GPU, loads, exports every
symbol include/mi_jpeg.h declares, and fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "mi_jpeg.h")).read()
    return sorted(set(re.findall(r"MIJ_API\s+[\w\s\*]+?\b(mij_\w+)\s*\(", hdr)))


def test_header_and_binding_agree(mij):
    from nvjpeg_imagecompressor_amd import _lib
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(mij):
    from nvjpeg_imagecompressor_amd import _lib
    L = _lib.load()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.mij_version().startswith(b"mi_jpeg")
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (\w+)", out))
    assert set(_declared()) <= exported
    # nothing but the C ABI leaks out of the library
    assert all(s.startswith("mij_") or s.startswith("_") for s in exported), exported


def test_library_has_gfx950_code_object(mij):
    from nvjpeg_imagecompressor_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"k_transform" in data and b"k_encode" in data


def test_no_hard_dependency_on_a_second_hip_runtime(mij):
    from nvjpeg_imagecompressor_amd import _lib
    out = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH], text=True)
    assert "libamdhip64" not in out     # bound at load time to the ONE runtime in the process (see build.py)


def test_fails_loudly_without_a_gpu(mij):
    from nvjpeg_imagecompressor_amd import _lib
    L = _lib.load()
    if L.mij_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mij.MiJpegError, match="no HIP device|no CPU fallback"):
        mij.Encoder(64, 64)
    r = mij.NvjpegCompressRunner(64, 64, verbose=False)
    with pytest.raises(mij.MiJpegError):
        r.buildCompressEnv()


def test_argument_validation_happens_before_device_use(mij):
    from nvjpeg_imagecompressor_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    for bad in (dict(width=0), dict(height=70000), dict(quality=0), dict(quality=101), dict(css=9)):
        kw = dict(width=64, height=64, quality=95, optimized_huffman=1, css=0, restart_interval=-1, device=0,
                  strip_mcu_row0=0, strip_mcu_rows=0)
        kw.update(bad)
        p = _lib.EncoderParams(**kw)
        assert L.mij_encoder_create(C.byref(p), C.byref(h)) == -1, bad     # MIJ_ERR_INVALID_ARG
        assert not h.value
    assert L.mij_encoder_create(None, C.byref(h)) == -1
    L.mij_encoder_destroy(None)   # NULL is a no-op


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nvjpeg_imagecompressor_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn
                assert "libjpeg_oracle" not in src and "mjo_" not in src, fn          # no link / dlopen / call
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", src), fn
