"""ctypes binding of the CPU oracle (oracle/jpeg_oracle.c).

TEST INFRASTRUCTURE ONLY. Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
the product package (nvjpeg_imagecompressor_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjpeg_oracle.so")

CSS_NAMES = {0: "444", 1: "422", 2: "420", 3: "440", 4: "411", 5: "410"}
CSS_FACTORS = {0: (1, 1), 1: (2, 1), 2: (2, 2), 3: (1, 2), 4: (4, 1), 5: (4, 2)}
PIXFMT = {"rgb": 0, "bgr": 1, "rgb_planar": 2, "bgr_planar": 3}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("jpeg_oracle.c", "jpeg_oracle_dec.c", "ijg_harness.c", "Makefile")]
    newest = max(os.path.getmtime(s) for s in srcs if os.path.exists(s))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i16p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_uint32)
        L.mjo_synth_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t]
        L.mjo_synth_rgb.restype = None
        L.mjo_quant_table.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mjo_quant_table.restype = None
        L.mjo_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.mjo_coefficients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.mjo_histogram.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.mjo_gen_optimal_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mjo_encode_coefficients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mjo_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mjo_free.argtypes = [C.c_void_p]
        L.mjo_free.restype = None
        if hasattr(L, "mjo_decode"):
            L.mjo_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int)]
            L.mjo_decode_info.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def synth_rgb(W, H, y0=0, rows=None):
    """SURVEY.md 8(d) synthetic RGB8 image rows [y0, y0+rows) of a W-wide image -> (rows, W, 3) uint8."""
    rows = H - y0 if rows is None else rows
    out = np.empty((rows, W, 3), np.uint8)
    lib().mjo_synth_rgb(out.ctypes.data, W, y0, rows, W * 3)
    return out


def synth_rgb_numpy(W, H, y0=0, rows=None):
    """Vectorised numpy restatement of the same generator (used to cross-check the C one and for big images)."""
    rows = H - y0 if rows is None else rows
    y = np.arange(y0, y0 + rows, dtype=np.int64)[:, None]
    x = np.arange(W, dtype=np.int64)[None, :]

    def tri(t, P):
        u = t % P
        return np.where(u < P // 2, u, P - 1 - u)

    gx, gy, gd = tri(x, 1024), tri(y, 768), tri(x + 2 * y, 320)
    base = np.stack([48 + gx * 96 // 512 + gy * 64 // 384 + 0 * x,
                     40 + gx * 64 // 512 + gd * 96 // 160,
                     56 + gy * 96 // 384 + gd * 48 // 160 + 0 * x], axis=-1)
    step = ((((x // 208) + (y // 250)) & 1) * 24)[..., None]
    idx = ((y * W + x)[..., None] * 3 + np.arange(3)).astype(np.uint64) & 0xFFFFFFFF
    h = ((idx * 0x9E3779B1) & 0xFFFFFFFF) ^ 0x4D493335
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    n = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)).astype(np.int64) - 510
    n >>= 4
    return np.clip(base + step + n, 0, 255).astype(np.uint8)


def rgb_to_ycc(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().mjo_rgb_to_ycc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib().mjo_rgb_to_ycc.restype = None
    lib().mjo_rgb_to_ycc(img.ctypes.data, img.shape[0] * img.shape[1], out.ctypes.data)
    return out


def quant_table(quality, which):
    out = np.empty(64, np.uint16)
    lib().mjo_quant_table(quality, which, out.ctypes.data)
    return out


def geometry(W, H, css):
    g = np.zeros(8, np.int32)
    if lib().mjo_geometry(W, H, css, g.ctypes.data):
        raise ValueError("bad geometry")
    return dict(hs=int(g[0]), vs=int(g[1]), mcux=int(g[2]), mcuy=int(g[3]), bpm=int(g[4]))


def _img_args(img, pixfmt):
    img = np.ascontiguousarray(img)
    fmt = PIXFMT[pixfmt]
    if fmt < 2:
        H, W, _ = img.shape
        stride = W * 3
    else:
        _, H, W = img.shape
        stride = W
    return img, W, H, stride, fmt


def coefficients(img, quality, css, pixfmt="rgb"):
    img, W, H, stride, fmt = _img_args(img, pixfmt)
    g = geometry(W, H, css)
    coef = np.empty((g["mcux"] * g["mcuy"], g["bpm"], 64), np.int16)
    rc = lib().mjo_coefficients(img.ctypes.data, W, H, stride, fmt, quality, css, coef.ctypes.data)
    if rc:
        raise RuntimeError("mjo_coefficients rc=%d" % rc)
    return coef


def histogram(coef, W, H, css, restart_interval):
    coef = np.ascontiguousarray(coef, np.int16)
    hist = np.zeros((4, 257), np.uint32)
    lib().mjo_histogram(coef.ctypes.data, W, H, css, restart_interval, hist.ctypes.data)
    return hist


def gen_optimal_table(freq):
    f = np.zeros(257, np.uint32)
    f[:len(freq)] = freq
    bits = np.zeros(17, np.uint8)
    vals = np.zeros(256, np.uint8)
    n = lib().mjo_gen_optimal_table(f.ctypes.data, bits.ctypes.data, vals.ctypes.data)
    if n < 0:
        raise RuntimeError("table generation failed")
    return bits, vals[:n]


def _take(pp, ln):
    out = C.string_at(pp.value, ln.value)
    lib().mjo_free(pp)
    return out


def encode_coefficients(coef, W, H, quality, css, optimize, restart_interval, headers=True, want_tables=False):
    coef = np.ascontiguousarray(coef, np.int16)
    pp, ln = C.c_void_p(), C.c_size_t()
    tabs = np.zeros((4, 273), np.uint8)
    rc = lib().mjo_encode_coefficients(coef.ctypes.data, W, H, quality, css, int(optimize), restart_interval,
                                       int(headers), tabs.ctypes.data, C.byref(pp), C.byref(ln))
    if rc:
        raise RuntimeError("mjo_encode_coefficients rc=%d" % rc)
    data = _take(pp, ln)
    return (data, tabs) if want_tables else data


def encode_strip(coef, W, strip_h, frame_h, quality, css, optimize, restart_interval, hist, rst_first, first, last):
    """Entropy-code one strip of MCU rows with the whole frame's statistics (see mjo_encode_strip)."""
    L = lib()
    L.mjo_encode_strip.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_long, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    coef = np.ascontiguousarray(coef, np.int16)
    hist = np.ascontiguousarray(hist, np.uint32)
    pp, ln = C.c_void_p(), C.c_size_t()
    rc = L.mjo_encode_strip(coef.ctypes.data, W, strip_h, frame_h, quality, css, int(optimize), restart_interval,
                            hist.ctypes.data, rst_first, (1 if first else 0) | (2 if last else 0), C.byref(pp), C.byref(ln))
    if rc:
        raise RuntimeError("mjo_encode_strip rc=%d" % rc)
    return _take(pp, ln)


def encode(img, quality=95, css=0, optimize=True, restart_interval=0, pixfmt="rgb"):
    img, W, H, stride, fmt = _img_args(img, pixfmt)
    pp, ln = C.c_void_p(), C.c_size_t()
    rc = lib().mjo_encode(img.ctypes.data, W, H, stride, fmt, quality, css, int(optimize), restart_interval,
                          C.byref(pp), C.byref(ln))
    if rc:
        raise RuntimeError("mjo_encode rc=%d" % rc)
    return _take(pp, ln)


def encode_progressive(img, quality=95, css=0, restart_interval=0, pixfmt="rgb"):
    """Progressive (SOF2) file with libjpeg's default scan script and one optimal table per scan."""
    img, W, H, stride, fmt = _img_args(img, pixfmt)
    pp, ln = C.c_void_p(), C.c_size_t()
    L = lib()
    L.mjo_encode_progressive.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    rc = L.mjo_encode_progressive(img.ctypes.data, W, H, stride, fmt, quality, css, restart_interval, C.byref(pp), C.byref(ln))
    if rc:
        raise RuntimeError("mjo_encode_progressive rc=%d" % rc)
    return _take(pp, ln)


def decode_info(jpg):
    info = np.zeros(8, np.int32)
    buf = np.frombuffer(jpg, np.uint8)
    rc = lib().mjo_decode_info(buf.ctypes.data, len(jpg), info.ctypes.data)
    if rc:
        raise RuntimeError("mjo_decode_info rc=%d" % rc)
    return dict(width=int(info[0]), height=int(info[1]), hs=int(info[2]), vs=int(info[3]),
                restart_interval=int(info[4]))


def decode(jpg, pixfmt="rgb"):
    """Decode a baseline JPEG (as libjpeg-turbo does by default: islow IDCT, fancy upsampling) -> (H, W, 3) uint8."""
    buf = np.frombuffer(jpg, np.uint8)
    inf = decode_info(jpg)
    out = np.empty((inf["height"], inf["width"], 3), np.uint8)
    w, h = C.c_int(), C.c_int()
    rc = lib().mjo_decode(buf.ctypes.data, len(jpg), PIXFMT[pixfmt], out.ctypes.data, inf["width"] * 3,
                          C.byref(w), C.byref(h))
    if rc:
        raise RuntimeError("mjo_decode rc=%d" % rc)
    return out


def psnr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
