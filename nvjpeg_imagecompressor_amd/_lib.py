"""ctypes binding of libmijpeg.so (the C ABI in include/mi_jpeg.h).

The product path. There is no CPU fallback here and nothing in this package imports oracle/: if the HIP library is
missing or no MI355X is visible, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIJ_LIB_PATH") or os.path.join(_HERE, "libmijpeg.so")   # override: experiment builds only

MIJ_OK = 0
MIJ_RESTART_AUTO = -1
CSS = {"444": 0, "422": 1, "420": 2, "440": 3, "411": 4, "410": 5}
INPUT_RGB, INPUT_BGR, INPUT_RGBI, INPUT_BGRI = 3, 4, 5, 6
NUM_STAGE_TIMES = 7
STAGE_NAMES = ("transform", "statistics", "tables", "entropy", "scan", "compact", "total")

# every symbol include/mi_jpeg.h declares (tests/test_abi.py checks the library exports all of them)
EXPORTS = (
    "mij_version", "mij_abi_version", "mij_source_hash", "mij_device_count", "mij_encoder_create", "mij_encoder_destroy", "mij_encoder_geometry",
    "mij_last_error", "mij_encode_device", "mij_encode_transform", "mij_encode_entropy", "mij_encode_tables", "mij_histogram_device",
    "mij_set_histogram_buffer", "mij_encode_result", "mij_retrieve_bitstream", "mij_encode_host",
    "mij_encoder_enable_timing", "mij_stage_times", "mij_debug_coefficients", "mij_debug_tables",
    "mij_synth_image_device", "mij_copy_bench_device", "mij_decoder_create", "mij_decoder_destroy", "mij_decoder_last_error", "mij_decode_info",
    "mij_decode_device", "mij_decode_sync", "mij_decode_host", "mij_residual_device", "mij_host_alloc", "mij_host_free",
    "mij_secondary_encode_host", "mij_secondary_decode_host", "mij_decode_last_ms", "mij_decoder_device",
    "mij_geometry_query", "mij_encode_entropy_sizes", "mij_encode_place", "mij_sharded_result", "mij_encoder_reserve_output",
    "mij_output_buffer", "mij_ipc_export", "mij_ipc_open", "mij_ipc_close", "mij_place_times", "mij_encode_residual_device",
    "mij_clock_probe_device", "mij_secondary_encode_host_ex", "mij_secondary_decode_host_ex", "mij_encode_residual_gain_device", "mij_residual_gain_device", "mij_output_is_uncached", "mij_decode_px_report",
    "mij_encode_prog_statistics", "mij_prog_histogram_buffer", "mij_encode_prog_emit", "mij_encode_prog_place",
)


class EncoderParams(C.Structure):
    """mij_encoder_params (include/mi_jpeg.h). struct_size is filled in by __init__: pass the other fields only."""
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int), ("height", C.c_int), ("quality", C.c_int), ("optimized_huffman", C.c_int),
                ("css", C.c_int), ("restart_interval", C.c_int), ("device", C.c_int),
                ("strip_mcu_row0", C.c_int), ("strip_mcu_rows", C.c_int), ("progressive", C.c_int)]

    def __init__(self, *args, **kw):
        if "struct_size" in kw:
            super().__init__(*args, **kw)
        else:
            super().__init__(C.sizeof(EncoderParams), *args, **kw)


class SecondaryParams(C.Structure):
    """mij_secondary_params (include/mi_jpeg.h): the second layer's own quality / sampling and the gain of the difference map."""
    _fields_ = [("struct_size", C.c_uint32), ("quality2", C.c_int), ("css2", C.c_int), ("gain", C.c_int)]

    def __init__(self, quality2=0, css2=-1, gain=1):
        super().__init__(C.sizeof(SecondaryParams), int(quality2), int(css2), int(gain))


class Geometry(C.Structure):
    _fields_ = [("hs", C.c_int), ("vs", C.c_int), ("mcu_w", C.c_int), ("mcu_h", C.c_int),
                ("mcus_per_row", C.c_int), ("mcu_rows", C.c_int), ("blocks_per_mcu", C.c_int),
                ("restart_interval", C.c_int), ("strip_first_mcu", C.c_int64), ("strip_mcus", C.c_int64),
                ("strip_y0", C.c_int), ("strip_rows", C.c_int)]


class Result(C.Structure):
    _fields_ = [("d_buffer", C.c_void_p), ("header_offset", C.c_size_t), ("header_bytes", C.c_size_t),
                ("scan_offset", C.c_size_t), ("scan_bytes", C.c_size_t), ("file_bytes", C.c_size_t)]


class MiJpegError(RuntimeError):
    pass


_lib = None
HIP_RUNTIME = None  # path of the HIP runtime libmijpeg.so was bound to


def _bind_hip_runtime():
    """libmijpeg.so carries no DT_NEEDED for libamdhip64 (build.py): make exactly ONE HIP runtime global before it
    loads. If PyTorch is already imported its bundled runtime is the one in the process (streams and device pointers
    then interoperate with torch); otherwise use the system ROCm runtime."""
    global HIP_RUNTIME
    import sys
    cands = []
    if os.environ.get("MIJ_HIP_RUNTIME"):
        cands.append(os.environ["MIJ_HIP_RUNTIME"])
    if "torch" in sys.modules:
        tl = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib")
        cands += [os.path.join(tl, "libamdhip64.so")]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so.7"), os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            HIP_RUNTIME = c
            return
        except OSError:
            continue
    raise MiJpegError("no HIP runtime (libamdhip64) found: mi_jpeg needs ROCm and an MI355X; there is no CPU fallback")


def load():
    """Load libmijpeg.so. Raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiJpegError("%s is missing: build it with `python -m nvjpeg_imagecompressor_amd.build` "
                          "(there is no CPU fallback)" % LIB_PATH)
    _bind_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.mij_version.restype = C.c_char_p
    L.mij_device_count.restype = C.c_int
    L.mij_last_error.restype = C.c_char_p
    L.mij_last_error.argtypes = [vp]
    L.mij_encoder_create.argtypes = [C.POINTER(EncoderParams), C.POINTER(vp)]
    L.mij_encoder_destroy.argtypes = [vp]
    L.mij_encoder_destroy.restype = None
    L.mij_encoder_geometry.argtypes = [vp, C.POINTER(Geometry)]
    L.mij_encode_device.argtypes = [vp, vp, sz, sz, C.c_int, vp]
    L.mij_encode_transform.argtypes = [vp, vp, sz, sz, C.c_int, vp]
    L.mij_encode_entropy.argtypes = [vp, vp]
    L.mij_encode_tables.argtypes = [vp, vp]
    L.mij_histogram_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.mij_set_histogram_buffer.argtypes = [vp, vp]
    L.mij_encode_result.argtypes = [vp, C.POINTER(Result)]
    L.mij_retrieve_bitstream.argtypes = [vp, vp, C.POINTER(sz)]
    L.mij_encode_host.argtypes = [vp, vp, sz, sz, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    L.mij_secondary_encode_host.argtypes = [vp, vp, vp, sz, sz, C.c_int, vp, C.POINTER(sz), vp, C.POINTER(sz)]
    L.mij_secondary_decode_host.argtypes = [vp, vp, sz, vp, sz, vp, sz, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    spp = C.POINTER(SecondaryParams)
    L.mij_secondary_encode_host_ex.argtypes = [vp, spp, vp, sz, sz, C.c_int, vp, C.POINTER(sz), vp, C.POINTER(sz)]
    L.mij_secondary_decode_host_ex.argtypes = [vp, spp, vp, sz, vp, sz, vp, sz, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mij_encode_residual_gain_device.argtypes = [vp, vp, sz, sz, C.c_int, vp, sz, sz, C.c_int, vp]
    L.mij_residual_gain_device.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, vp]
    L.mij_output_is_uncached.argtypes = [vp]
    L.mij_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.mij_host_free.argtypes = [vp]
    L.mij_host_free.restype = None
    L.mij_encoder_enable_timing.argtypes = [vp, C.c_int]
    L.mij_stage_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.mij_debug_coefficients.argtypes = [vp, vp, sz]
    L.mij_debug_tables.argtypes = [vp, vp]
    L.mij_copy_bench_device.argtypes = [vp, vp, sz, vp]
    L.mij_synth_image_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, sz, C.c_int, vp]
    dp = C.POINTER(C.c_double)
    L.mij_clock_probe_device.argtypes = [C.c_int, vp, dp, dp, dp]
    L.mij_decoder_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mij_decoder_destroy.argtypes = [vp]
    L.mij_decoder_destroy.restype = None
    L.mij_decoder_last_error.argtypes = [vp]
    L.mij_decoder_last_error.restype = C.c_char_p
    ip = C.POINTER(C.c_int)
    L.mij_decode_info.argtypes = [vp, sz, ip, ip, ip, ip]
    L.mij_decode_px_report.argtypes = [vp, ip, ip]
    L.mij_decode_device.argtypes = [vp, vp, sz, vp, sz, sz, C.c_int, vp]
    L.mij_decode_sync.argtypes = [vp, C.POINTER(C.c_float)]
    L.mij_decode_host.argtypes = [vp, vp, sz, vp, sz, C.c_int, ip, ip]
    L.mij_residual_device.argtypes = [vp, vp, vp, sz, C.c_int, vp]
    L.mij_geometry_query.argtypes = [C.POINTER(EncoderParams), C.POINTER(Geometry)]
    L.mij_encode_prog_statistics.argtypes = [vp, vp]
    L.mij_prog_histogram_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.mij_encode_prog_emit.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mij_encode_prog_place.argtypes = [vp, C.POINTER(C.c_uint64), vp, sz, C.c_uint64, C.c_int, vp]
    L.mij_encode_entropy_sizes.argtypes = [vp, vp, vp]
    L.mij_encode_place.argtypes = [vp, vp, sz, vp, C.c_int, C.c_int, vp]
    L.mij_sharded_result.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(Result)]
    L.mij_place_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.mij_encode_residual_device.argtypes = [vp, vp, sz, sz, C.c_int, vp, sz, sz, vp]
    L.mij_encoder_reserve_output.argtypes = [vp, sz]
    L.mij_output_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.mij_ipc_export.argtypes = [vp, vp]
    L.mij_ipc_open.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.mij_ipc_close.argtypes = [vp]
    L.mij_decode_last_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.mij_decoder_device.argtypes = [vp]
    L.mij_abi_version.restype = C.c_int
    L.mij_source_hash.restype = C.c_char_p
    _lib = L
    return L


def check(rc, handle=None, what="mi_jpeg call"):
    if rc != MIJ_OK:
        msg = load().mij_last_error(handle)
        raise MiJpegError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
