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


def test_shipped_library_matches_the_sources(mij):
    """The build is stale by CONTENT, not mtime: the hash compiled into the .so is the hash of the tree."""
    from nvjpeg_imagecompressor_amd import _lib, build as B
    L = _lib.load()
    assert L.mij_source_hash().decode() == B.source_hash()
    assert not B.needs_build()
    assert L.mij_abi_version() == 2


def test_encoder_params_carry_their_size(mij):
    """A caller compiled against another layout is refused instead of being read past its end (ABI 1 had no struct_size and
    nine ints; a tenth field was then added at the end)."""
    from nvjpeg_imagecompressor_amd import _lib
    L = _lib.load()
    h = C.c_void_p()

    class OldNineInts(C.Structure):      # the ABI-1 layout INTEGRATION.md used to show
        _fields_ = [(n, C.c_int) for n in ("width", "height", "quality", "optimized_huffman", "css", "restart_interval", "device",
                                            "strip_mcu_row0", "strip_mcu_rows")]
    assert L.mij_encoder_create(C.cast(C.byref(OldNineInts(64, 64, 95, 1, 0, -1, 0, 0, 0)), C.POINTER(_lib.EncoderParams)), C.byref(h)) == -1
    assert b"struct_size" in L.mij_last_error(None)
    p = _lib.EncoderParams(64, 64, 95, 1, 0, -1, 0, 0, 0, 0)
    assert p.struct_size == C.sizeof(_lib.EncoderParams) == 44
    p.struct_size = 48          # a newer caller than the library
    assert L.mij_encoder_create(C.byref(p), C.byref(h)) == -1
    p.struct_size = 40          # older caller without `progressive`: accepted (fails later only for want of a device here)
    rc = L.mij_encoder_create(C.byref(p), C.byref(h))
    assert rc in (0, -3), rc
    if rc == 0:
        L.mij_encoder_destroy(h)


# ---- header fuzz (host-side parser only: mij_decode_info needs no device) -------------------------------------------
def _seg(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload


def _dht(tc_th, counts):
    n = sum(counts)
    return _seg(0xC4, bytes([tc_th]) + bytes(counts) + bytes(i & 255 for i in range(n)))


def _sof(marker=0xC0, w=16, h=16, comps=((1, 0x11, 0), (2, 0x11, 1), (3, 0x11, 1))):
    return _seg(marker, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([len(comps)]) +
                b"".join(bytes(c) for c in comps))


def _sos(ids=(1, 2, 3), ss=0, se=63, ahal=0):
    return _seg(0xDA, bytes([len(ids)]) + b"".join(bytes([i, 0x00 if i == 1 else 0x11]) for i in ids) + bytes([ss, se, ahal]))


_GOOD_COUNTS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]     # Annex K DC luminance


def _info(mij, data):
    from nvjpeg_imagecompressor_amd import _lib
    L = _lib.load()
    w, h, css, ri = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = L.mij_decode_info(data, len(data), C.byref(w), C.byref(h), C.byref(css), C.byref(ri))
    return rc, (w.value, h.value)


def test_header_fuzz_is_rejected_cleanly(mij):
    """Crafted headers must come back as MIJ_ERR_BAD_STREAM (-6), never crash: the oversubscribed DHT used to run
    `look[code << (9 - l)]` far past the table (stack overflow in mij_decode_info / mij_decode_device)."""
    soi, eoi = b"\xff\xd8", b"\xff\xd9"
    tables = b"".join(_dht(t, _GOOD_COUNTS) for t in (0x00, 0x10, 0x01, 0x11))
    dqt = _seg(0xDB, bytes([0]) + bytes([1] * 64)) + _seg(0xDB, bytes([1]) + bytes([1] * 64))
    good = soi + dqt + _sof() + tables + _sos() + b"\x00" * 8 + eoi
    assert _info(mij, good) == (0, (16, 16))
    bad = {
        "255 codes of length 1": soi + _dht(0x00, [255] + [0] * 15) + eoi,
        "3 codes of length 1": soi + _dht(0x00, [3] + [0] * 15) + eoi,
        "oversubscribed at length 9": soi + _dht(0x00, [1, 1, 1, 1, 1, 1, 1, 1, 3] + [0] * 7) + eoi,
        "oversubscribed at length 16": soi + _dht(0x10, [1] * 15 + [200]) + eoi,
        "empty table": soi + _dht(0x00, [0] * 16) + eoi,
        "more than 256 values": soi + _seg(0xC4, bytes([0x00]) + bytes([0, 0, 0, 0, 0, 0, 0, 0, 255, 255] + [0] * 6) + bytes(510)) + eoi,
        "table id 2": soi + _dht(0x02, _GOOD_COUNTS) + eoi,
        "DHT values truncated": soi + _seg(0xC4, bytes([0x00]) + bytes(_GOOD_COUNTS) + bytes(3)) + eoi,
        "truncated DQT segment": soi + b"\xff\xdb\x00\x43" + bytes(10),
        "16-bit DQT": soi + _seg(0xDB, bytes([0x10]) + bytes(128)) + _sof() + tables + _sos() + eoi,
        "truncated SOF": soi + b"\xff\xc0\x00\x11\x08\x00\x10",
        "SOF shorter than its component count": soi + _seg(0xC0, bytes([8, 0, 16, 0, 16, 3, 1, 0x11, 0])) + eoi,
        "12-bit SOF": soi + _seg(0xC0, bytes([12, 0, 16, 0, 16, 1, 1, 0x11, 0])) + eoi,
        "zero-size frame": soi + dqt + _sof(w=0) + tables + _sos() + eoi,
        "2 components": soi + dqt + _sof(comps=((1, 0x11, 0), (2, 0x11, 1))) + tables + _sos((1, 2)) + eoi,
        "subsampled chroma": soi + dqt + _sof(comps=((1, 0x11, 0), (2, 0x21, 1), (3, 0x11, 1))) + tables + _sos() + eoi,
        "3x1 luma": soi + dqt + _sof(comps=((1, 0x31, 0), (2, 0x11, 1), (3, 0x11, 1))) + tables + _sos() + eoi,
        "duplicate SOF": soi + dqt + _sof() + _sof(w=8, h=8) + tables + _sos() + eoi,
        "second SOF after a scan": soi + dqt + _sof(0xC2, 4096, 4096) + tables + _sos((1, 2, 3), 0, 0, 0) + b"\x00" * 4 +
                                   _sof(0xC2, 8, 8) + _sos((1,), 1, 63, 0) + eoi,
        "SOF after a scan changes the component count": soi + dqt + _sof(0xC2) + tables + _sos((1, 2, 3), 0, 0, 0) + b"\x00" * 4 +
                                   _sof(0xC2, comps=((1, 0x11, 0),)) + _sos((2,), 1, 63, 0) + eoi,
        "SOS before SOF": soi + tables + _sos() + eoi,
        "truncated SOS": soi + dqt + _sof() + tables + b"\xff\xda\x00\x0c\x03\x01",
        "SOS with an unknown component": soi + dqt + _sof() + tables + _sos((1, 2, 9)) + eoi,
        "SOS without its tables": soi + dqt + _sof() + _dht(0x00, _GOOD_COUNTS) + _sos() + eoi,
        "sequential scan with a spectral band": soi + dqt + _sof() + tables + _sos((1, 2, 3), 1, 5, 0) + eoi,
        "progressive band reversed": soi + dqt + _sof(0xC2) + tables + _sos((1,), 9, 3, 0) + eoi,
        "arithmetic coding": soi + dqt + _sof(0xC9) + tables + _sos() + eoi,
        "no scan": soi + dqt + _sof() + tables + eoi,
        "no SOI": b"\x00\x00" + dqt,
        "marker expected": soi + b"\x12\x34\x56\x78",
    }
    for name, data in bad.items():
        rc, _ = _info(mij, data)
        assert rc == -6, (name, rc)


def test_geometry_query_needs_no_device(mij):
    """mij_geometry_query is pure arithmetic: a sharding host cuts its strips with it before any handle exists. It must agree
    with the oracle's geometry and validate like mij_encoder_create."""
    from oracle import oracle as O
    for (W, H, css) in [(8320, 40000, 1), (8320, 40000, 2), (208, 250, 1), (333, 77, 0), (17, 33, 4), (64, 64, 5)]:
        g = mij.geometry_query(W, H, 95, True, css)
        og = O.geometry(W, H, css)
        assert (g["mcus_per_row"], g["mcu_rows"], g["hs"], g["vs"]) == (og["mcux"], og["mcuy"], og["hs"], og["vs"])
        assert g["strip_first_mcu"] == 0 and g["strip_mcus"] == og["mcux"] * og["mcuy"] and g["strip_rows"] == H
        assert 1 <= g["restart_interval"] <= 65535
    assert mij.geometry_query(8320, 40000, 95, True, 1)["restart_interval"] == 64         # the headline configuration: 256 blocks = 4 full batches
    for css, bpm in ((0, 3), (1, 4), (2, 6), (3, 4), (4, 6), (5, 10)):                     # AUTO: every interval a whole number of 64-block batches
        assert mij.geometry_query(8320, 40000, 95, True, css)["restart_interval"] * bpm % 64 == 0
    assert mij.geometry_query(8320, 40000, 95, True, 1, 112)["restart_interval"] == 112   # an interval that does not divide the row: still fine
    with pytest.raises(mij.MiJpegError):
        mij.geometry_query(0, 10)
    with pytest.raises(mij.MiJpegError):
        mij.geometry_query(64, 64, 95, True, 9)
