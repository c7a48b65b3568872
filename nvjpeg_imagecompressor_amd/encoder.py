"""Python host layer over the C ABI: an Encoder handle object plus a mirror of the reference's
`NvjpegCompressRunner` class surface (reference src/ImageCompressorDll/ImageCompressor.h:22-42) for hosts that drive
the path from Python. numpy arrays stand in for cv::Mat (H x W x 3 uint8, BGR, as cv::imread returns).

Nothing here computes JPEG arithmetic: every call goes to libmijpeg.so (HIP). No CPU fallback.
"""
import ctypes as C
import time

import numpy as np

from . import _lib
from ._lib import CSS, INPUT_BGR, INPUT_BGRI, INPUT_RGB, INPUT_RGBI, MIJ_RESTART_AUTO, MiJpegError

_FMT = {"rgb": INPUT_RGBI, "bgr": INPUT_BGRI, "rgb_planar": INPUT_RGB, "bgr_planar": INPUT_BGR}


def _css_value(css):
    if isinstance(css, str):
        return CSS[css.replace(":", "")]
    return int(css)


class Encoder:
    """One encoder state + device workspace (initCompressEnv / destoryCompressEnv, reference ImageCompressorImpl.cu:19-65)."""

    def __init__(self, width, height, quality=95, optimized_huffman=True, css=0, restart_interval=MIJ_RESTART_AUTO,
                 device=0, strip_mcu_row0=0, strip_mcu_rows=0, progressive=False):
        self._L = _lib.load()
        self._h = C.c_void_p()
        p = _lib.EncoderParams(width, height, quality, int(bool(optimized_huffman)), _css_value(css),
                               restart_interval, device, strip_mcu_row0, strip_mcu_rows, int(bool(progressive)))
        rc = self._L.mij_encoder_create(C.byref(p), C.byref(self._h))
        if rc:
            msg = self._L.mij_last_error(None)
            self._h = C.c_void_p()
            raise MiJpegError("mij_encoder_create failed (rc=%d): %s" % (rc, msg.decode() if msg else "?"))
        g = _lib.Geometry()
        _lib.check(self._L.mij_encoder_geometry(self._h, C.byref(g)), self._h)
        self.geometry = {f[0]: getattr(g, f[0]) for f in g._fields_}
        self.width, self.height = width, height

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.mij_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- device-resident path (the timed path) -------------------------------------------------------------------
    def encode_device(self, d_ptr, pitch, fmt="bgr", plane_stride=0, stream=0):
        _lib.check(self._L.mij_encode_device(self._h, C.c_void_p(d_ptr), pitch, plane_stride, _FMT[fmt],
                                             C.c_void_p(stream)), self._h, "mij_encode_device")

    def transform(self, d_ptr, pitch, fmt="bgr", plane_stride=0, stream=0):
        _lib.check(self._L.mij_encode_transform(self._h, C.c_void_p(d_ptr), pitch, plane_stride, _FMT[fmt],
                                                C.c_void_p(stream)), self._h, "mij_encode_transform")

    def tables(self, stream=0):
        """Optional: build the tables on `stream` (possibly another than the transform's); see mi_jpeg.h."""
        _lib.check(self._L.mij_encode_tables(self._h, C.c_void_p(stream)), self._h, "mij_encode_tables")

    def entropy(self, stream=0):
        _lib.check(self._L.mij_encode_entropy(self._h, C.c_void_p(stream)), self._h, "mij_encode_entropy")

    # -- strip sharding with sizes / offsets kept on the device (mi_jpeg.h "Strip sharding") ------------------------
    def entropy_sizes(self, d_size_slot, stream=0):
        _lib.check(self._L.mij_encode_entropy_sizes(self._h, C.c_void_p(d_size_slot), C.c_void_p(stream)), self._h,
                   "mij_encode_entropy_sizes")

    def place(self, d_file_scan, file_capacity, d_sizes, rank, world, stream=0):
        _lib.check(self._L.mij_encode_place(self._h, C.c_void_p(d_file_scan), file_capacity, C.c_void_p(d_sizes), rank, world,
                                            C.c_void_p(stream)), self._h, "mij_encode_place")

    def sharded_result(self, d_sizes, rank, world):
        r = _lib.Result()
        _lib.check(self._L.mij_sharded_result(self._h, C.c_void_p(d_sizes), rank, world, C.byref(r)), self._h, "mij_sharded_result")
        return {f[0]: getattr(r, f[0]) for f in r._fields_}

    def place_times(self):
        """(ms of stuffing + compaction, ms of the put into the root's buffer) of the last `place` issued with timing on."""
        ms = (C.c_float * 2)()
        _lib.check(self._L.mij_place_times(self._h, ms), self._h, "mij_place_times")
        return float(ms[0]), float(ms[1])

    # -- progressive output in strips (mi_jpeg.h "Progressive output in STRIPS") ---------------------------------------
    PROG_SCANS, PROG_PLACE_HEADERS, PROG_PLACE_EOI = 10, 1, 2

    def prog_statistics(self, stream=0):
        _lib.check(self._L.mij_encode_prog_statistics(self._h, C.c_void_p(stream)), self._h, "mij_encode_prog_statistics")

    def prog_histogram_buffer(self):
        """(device pointer, uint32 words) of the ten scans' statistics: what the ranks of a sharded encode sum in place."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self._L.mij_prog_histogram_buffer(self._h, C.byref(p), C.byref(n)), self._h, "mij_prog_histogram_buffer")
        return p.value, n.value

    def prog_emit(self, stream=0):
        """-> (sizes[10], header_bytes[10]) on the host (waits for the ten scans)."""
        a, b = (C.c_uint64 * 10)(), (C.c_uint64 * 10)()
        _lib.check(self._L.mij_encode_prog_emit(self._h, C.c_void_p(stream), a, b), self._h, "mij_encode_prog_emit")
        return [int(v) for v in a], [int(v) for v in b]

    def prog_place(self, offsets, d_file=0, capacity=0, file_bytes=0, flags=0, stream=0):
        o = (C.c_uint64 * 10)(*[int(v) for v in offsets])
        _lib.check(self._L.mij_encode_prog_place(self._h, o, C.c_void_p(d_file or 0), capacity, file_bytes, flags, C.c_void_p(stream)), self._h,
                   "mij_encode_prog_place")

    def residual_device(self, d_src, pitch, d_dst, fmt="bgr", plane_stride=0, dst_pitch=None, dst_plane_stride=None, stream=0, gain=1):
        """After encode_device / transform of image I on this handle: d_dst <- clip((I - D) * gain + 128) with D = what a decoder
        makes of this handle's file, computed from the coefficients (d_src = I); d_src = 0 / None: d_dst <- D itself."""
        _lib.check(self._L.mij_encode_residual_gain_device(self._h, C.c_void_p(d_src or 0), pitch, plane_stride, _FMT[fmt], C.c_void_p(d_dst),
                                                           pitch if dst_pitch is None else dst_pitch,
                                                           plane_stride if dst_plane_stride is None else dst_plane_stride, int(gain),
                                                           C.c_void_p(stream)),
                   self._h, "mij_encode_residual_gain_device")

    def reserve_output(self, scan_capacity):
        _lib.check(self._L.mij_encoder_reserve_output(self._h, scan_capacity), self._h, "mij_encoder_reserve_output")

    def output_is_uncached(self):
        """True when reserve_output gave this handle a device-uncached output buffer (the sharded path: peers write into it)."""
        return bool(self._L.mij_output_is_uncached(self._h))

    def output_buffer(self):
        """(device pointer of the output buffer, offset of the scan area inside it, capacity of the scan area)."""
        p, off, cap = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _lib.check(self._L.mij_output_buffer(self._h, C.byref(p), C.byref(off), C.byref(cap)), self._h)
        return p.value, off.value, cap.value

    def set_histogram_buffer(self, d_ptr):
        _lib.check(self._L.mij_set_histogram_buffer(self._h, C.c_void_p(d_ptr)), self._h)

    def result(self):
        r = _lib.Result()
        _lib.check(self._L.mij_encode_result(self._h, C.byref(r)), self._h, "mij_encode_result")
        return {f[0]: getattr(r, f[0]) for f in r._fields_}

    def retrieve(self):
        """nvjpegEncodeRetrieveBitstream protocol (reference .cu:285-287): size query, then copy."""
        n = C.c_size_t(0)
        _lib.check(self._L.mij_retrieve_bitstream(self._h, None, C.byref(n)), self._h, "mij_retrieve_bitstream")
        buf = np.empty(n.value, np.uint8)
        _lib.check(self._L.mij_retrieve_bitstream(self._h, buf.ctypes.data, C.byref(n)), self._h)
        return buf[:n.value].tobytes()

    # -- host path (CompressWorker, reference .cu:269-294) --------------------------------------------------------
    def encode_host(self, img, fmt="bgr", as_view=False):
        """Host image in, JFIF bytes out. `as_view=True` returns a numpy view of the library's page-locked output buffer
        (valid until the next call on this encoder) instead of copying it into a bytes object."""
        img = np.ascontiguousarray(img, np.uint8)
        if fmt in ("rgb", "bgr"):
            pitch, plane = img.shape[1] * 3, 0
        else:
            pitch, plane = img.shape[2], img.shape[1] * img.shape[2]
        out, n = C.c_void_p(), C.c_size_t()
        _lib.check(self._L.mij_encode_host(self._h, img.ctypes.data, pitch, plane, _FMT[fmt], C.byref(out), C.byref(n)),
                   self._h, "mij_encode_host")
        if as_view:
            return np.ctypeslib.as_array(C.cast(out.value, C.POINTER(C.c_uint8)), shape=(n.value,))
        return C.string_at(out.value, n.value)

    # -- instrumentation -------------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        _lib.check(self._L.mij_encoder_enable_timing(self._h, int(on)), self._h)

    def stage_times(self):
        ms = (C.c_float * _lib.NUM_STAGE_TIMES)()
        _lib.check(self._L.mij_stage_times(self._h, ms), self._h)
        return dict(zip(_lib.STAGE_NAMES, [float(v) for v in ms]))

    def debug_coefficients(self):
        g = self.geometry
        out = np.empty((g["strip_mcus"], g["blocks_per_mcu"], 64), np.int16)
        _lib.check(self._L.mij_debug_coefficients(self._h, out.ctypes.data, out.size), self._h)
        return out

    def debug_tables(self):
        out = np.zeros((4, 273), np.uint8)
        _lib.check(self._L.mij_debug_tables(self._h, out.ctypes.data), self._h)
        return out


class Decoder:
    """Decoder state + device workspace (initDecodeEnv / destoryDecodeEnv, reference ImageCompressorImpl.cu:67-117)."""

    def __init__(self, device=0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.mij_decoder_create(device, C.byref(self._h))
        if rc:
            msg = self._L.mij_decoder_last_error(None)
            self._h = C.c_void_p()
            raise MiJpegError("mij_decoder_create failed (rc=%d): %s" % (rc, msg.decode() if msg else "?"))

    def _check(self, rc, what):
        if rc:
            msg = self._L.mij_decoder_last_error(self._h)
            raise MiJpegError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.mij_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def info(jpeg):
        L = _lib.load()
        w, h, css, ri = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        buf = np.frombuffer(jpeg, np.uint8)
        rc = L.mij_decode_info(buf.ctypes.data, len(jpeg), C.byref(w), C.byref(h), C.byref(css), C.byref(ri))
        if rc:
            msg = L.mij_decoder_last_error(None)
            raise MiJpegError("mij_decode_info failed (rc=%d): %s" % (rc, msg.decode() if msg else "?"))
        return dict(width=w.value, height=h.value, css=css.value, restart_interval=ri.value)

    def decode_host(self, jpeg, fmt="bgr"):
        """JPEG bytes -> H x W x 3 uint8 (interleaved) or 3 x H x W (planar), like DecodeWorker + getCVImageOnCPU."""
        inf = self.info(jpeg)
        buf = np.frombuffer(jpeg, np.uint8)
        planar = fmt.endswith("_planar")
        out = np.empty((3, inf["height"], inf["width"]) if planar else (inf["height"], inf["width"], 3), np.uint8)
        w, h = C.c_int(), C.c_int()
        self._check(self._L.mij_decode_host(self._h, buf.ctypes.data, len(jpeg), out.ctypes.data,
                                            inf["width"] * (1 if planar else 3), _FMT[fmt], C.byref(w), C.byref(h)), "mij_decode_host")
        return out

    def decode_device(self, jpeg, d_ptr, pitch, fmt="bgr", plane_stride=0, stream=0):
        buf = np.frombuffer(jpeg, np.uint8)
        self._check(self._L.mij_decode_device(self._h, buf.ctypes.data, len(jpeg), C.c_void_p(d_ptr), pitch, plane_stride,
                                              _FMT[fmt], C.c_void_p(stream)), "mij_decode_device")

    def decode_device_ptr(self, d_jpeg, nbytes, d_ptr, pitch, fmt="bgr", plane_stride=0, stream=0):
        """Like decode_device for a file that already sits in device memory (e.g. Encoder.result()): only the header is
        copied to the host for parsing, the entropy-coded data is decoded in place."""
        self._check(self._L.mij_decode_device(self._h, C.c_void_p(d_jpeg), nbytes, C.c_void_p(d_ptr), pitch, plane_stride,
                                              _FMT[fmt], C.c_void_p(stream)), "mij_decode_device")

    def px_report(self):
        """(scans the parallel progressive decoder was tried on, scans it decoded) for the last decode (mij_decode_px_report)."""
        a, b = C.c_int(), C.c_int()
        self._check(self._L.mij_decode_px_report(self._h, C.byref(a), C.byref(b)), "mij_decode_px_report")
        return a.value, b.value

    def last_ms(self):
        ms = C.c_float()
        self._check(self._L.mij_decode_last_ms(self._h, C.byref(ms)), "mij_decode_last_ms")
        return float(ms.value)

    def sync(self):
        ms = C.c_float()
        self._check(self._L.mij_decode_sync(self._h, C.byref(ms)), "mij_decode_sync")
        return float(ms.value)


def geometry_query(width, height, quality=95, optimized_huffman=True, css=0, restart_interval=MIJ_RESTART_AUTO, progressive=False):
    """Geometry of the WHOLE image from the parameters alone (mij_geometry_query: no device, nothing allocated)."""
    L = _lib.load()
    p = _lib.EncoderParams(width, height, quality, int(bool(optimized_huffman)), _css_value(css), restart_interval, 0, 0, 0,
                           int(bool(progressive)))
    g = _lib.Geometry()
    rc = L.mij_geometry_query(C.byref(p), C.byref(g))
    if rc:
        msg = L.mij_last_error(None)
        raise MiJpegError("mij_geometry_query failed (rc=%d): %s" % (rc, msg.decode() if msg else "?"))
    return {f[0]: getattr(g, f[0]) for f in g._fields_}


def ipc_export(d_ptr):
    """64 opaque bytes that let another process map the allocation containing `d_ptr` (hipIpcGetMemHandle)."""
    L = _lib.load()
    h = (C.c_uint8 * 64)()
    _lib.check(L.mij_ipc_export(C.c_void_p(d_ptr), h), None, "mij_ipc_export")
    return bytes(h)


def ipc_open(device, handle):
    L = _lib.load()
    p = C.c_void_p()
    buf = (C.c_uint8 * 64).from_buffer_copy(handle)
    _lib.check(L.mij_ipc_open(device, buf, C.byref(p)), None, "mij_ipc_open")
    return p.value


def ipc_close(d_ptr):
    _lib.load().mij_ipc_close(C.c_void_p(d_ptr))


def residual_device(d_a, d_b, d_out, nbytes, mode, stream=0, gain=1):
    """mode -1: out = clip((a - b) * gain + 128); mode +1: out = clip(a + (b - 128) / gain, halves up) (difference-map compression,
    SURVEY.md 8a A9; gain: mij_secondary_params)."""
    L = _lib.load()
    _lib.check(L.mij_residual_gain_device(C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out), nbytes, mode, int(gain), C.c_void_p(stream)), None,
               "mij_residual_gain_device")


class _PinnedBlock:
    def __init__(self, L, ptr):
        self._L, self.ptr = L, ptr

    def __del__(self):
        try:
            self._L.mij_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (mij_host_alloc): uploads from it are direct DMA, not staged copies."""
    L = _lib.load()
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = C.c_void_p()
    _lib.check(L.mij_host_alloc(C.byref(ptr), max(nbytes, 1)), None, "mij_host_alloc")
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(ptr.value)
    buf._mij_owner = _PinnedBlock(L, ptr)    # freed when the last array over `buf` is gone
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def library_source_hash():
    """SHA-256 (hex) of the sources the LOADED libmijpeg.so was compiled from (mij_source_hash): what build.source_hash()
    gives for the tree it belongs to."""
    return _lib.load().mij_source_hash().decode()


def copy_bench_device(d_dst, d_src, nbytes, stream=0):
    """Streaming-copy yardstick kernel (bench only): 16 B per lane, grid-stride."""
    L = _lib.load()
    _lib.check(L.mij_copy_bench_device(C.c_void_p(d_dst), C.c_void_p(d_src), nbytes, C.c_void_p(stream)), None, "mij_copy_bench_device")


def clock_probe_device(iters=1024, stream=0):
    """Effective shader clock (bench only): {"valu_mhz": from the issue rate of a fixed vector-ALU loop, "counter_mhz": from the
    shader-clock / constant-rate counter ratio inside the same launch (0.0 if unknown), "launch_ms"}."""
    L = _lib.load()
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _lib.check(L.mij_clock_probe_device(int(iters), C.c_void_p(stream), C.byref(a), C.byref(b), C.byref(c)), None, "mij_clock_probe_device")
    return {"valu_mhz": a.value, "counter_mhz": b.value, "launch_ms": c.value}


def synth_image_device(d_ptr, width, y0, rows, pitch, bgr=False, stream=0):
    L = _lib.load()
    _lib.check(L.mij_synth_image_device(C.c_void_p(d_ptr), width, y0, rows, pitch, int(bgr), C.c_void_p(stream)),
               None, "mij_synth_image_device")


class NvjpegCompressRunner:
    """Mirror of the reference facade (ImageCompressor.h:22-42, ImageCompressor.cpp:14-101): same method names,
    argument meaning and failure convention (empty result + run_state 0; nothing ever exit()s).

    Additive extensions: css / restart_interval constructor keywords, in-memory decode.
    `run_state` is returned as the second element of a tuple because Python has no out-parameters.
    """

    def __init__(self, width=8320, height=40000, quality=95, optimize=True, css=0, restart_interval=MIJ_RESTART_AUTO,
                 device=0, verbose=True, progressive=False):
        self.width, self.height, self.quality, self.optimize = width, height, quality, optimize
        self.css, self.restart_interval, self.device, self.verbose = css, restart_interval, device, verbose
        self.progressive = progressive   # the reference's nvJPEG encoding (ImageCompressorImpl.cu:28); default: baseline, the fast path
        self._enc = None
        self._dec = None

    def buildCompressEnv(self):
        if self._enc is None:  # a second build is a no-op (the reference leaks here)
            self._enc = Encoder(self.width, self.height, self.quality, self.optimize, self.css, self.restart_interval,
                                self.device, progressive=self.progressive)
            self._enc.enable_timing(True)

    def deleteCompressEnv(self):
        if self._enc is not None:
            self._enc.close()
            self._enc = None

    def buildDecodeEnv(self):
        if self._dec is None:
            self._dec = Decoder(self.device)

    def deleteDecodeEnv(self):
        if self._dec is not None:
            self._dec.close()
            self._dec = None

    def decode(self, image_path):
        """Path of a JPEG file -> (H x W x 3 uint8 BGR, run_state); (None, 0) on failure (reference ImageCompressor.cpp:63-88)."""
        t0 = time.perf_counter()
        result = None
        try:
            try:
                data = open(image_path, "rb").read()
            except OSError:
                print("Failed to open JPEG file.")
                return None, 0
            if self._dec is None:
                raise MiJpegError("decode() before buildDecodeEnv()")
            result = self._dec.decode_host(data, "bgr")
            if self.verbose:
                print("=> Decode Cost time : %gms" % self._dec.last_ms())   # reference ImageCompressorImpl.cu:373
        except MiJpegError as e:
            print("[ERROR] Exception caught : %s" % e)
            result = None
        if self.verbose:
            print("[INFO] NvjpegCompressRunner Decode Func Cost Time : %d ms" % int((time.perf_counter() - t0) * 1e3))
        return result, (0 if result is None else 1)

    def compress(self, image):
        """image: H x W x 3 uint8 BGR (cv::Mat CV_8UC3). Returns (bytes, run_state)."""
        t0 = time.perf_counter()
        out = b""
        try:
            if self._enc is None:
                raise MiJpegError("compress() before buildCompressEnv()")
            image = np.asarray(image)
            if image.dtype != np.uint8 or image.ndim != 3 or image.shape != (self.height, self.width, 3):
                raise MiJpegError("image must be uint8 %dx%dx3, got %s %s" % (self.height, self.width, image.dtype,
                                                                               image.shape))
            out = self._enc.encode_host(image, "bgr")
            if self.verbose:
                print("=> Compress Cost time : %gms" % self._enc.stage_times()["total"])
        except MiJpegError as e:
            print("[ERROR] Exception caught: %s" % e)
            out = b""
        if self.verbose:
            print("[INFO] NvjpegCompressRunner Compress Func Cost Time : %d ms" % int((time.perf_counter() - t0) * 1e3))
        return out, (0 if not out else 1)

    def secondaryCompress(self, image, quality2=None, css2=None, gain=1):
        """Secondary ("difference map") compression, reference README.md:8: returns (primary, secondary, run_state) where
        primary = JPEG(image) and secondary = JPEG(clip((image - decode(primary)) * gain + 128)) coded at `quality2` / `css2`
        (default: the first layer's; mij_secondary_params in mi_jpeg.h). Needs the compress environment only: decode(primary)
        comes from the encoder's own coefficients."""
        try:
            if self._enc is None:
                raise MiJpegError("secondaryCompress() before buildCompressEnv()")
            image = np.ascontiguousarray(image, np.uint8)
            if image.shape != (self.height, self.width, 3):
                raise MiJpegError("image must be uint8 %dx%dx3" % (self.height, self.width))
            sp = _lib.SecondaryParams(0 if quality2 is None else quality2, -1 if css2 is None else _css_value(css2), gain)
            # a residual image is close to noise: a layer can exceed the raw size (MIJ_ERR_OVERFLOW reports what is needed)
            cap1 = cap2 = image.size // 2 + 65536
            L = self._enc._L
            for _ in range(4):
                b1, b2 = np.empty(cap1, np.uint8), np.empty(cap2, np.uint8)
                n1, n2 = C.c_size_t(cap1), C.c_size_t(cap2)
                rc = L.mij_secondary_encode_host_ex(self._enc._h, C.byref(sp), image.ctypes.data, self.width * 3, 0, _FMT["bgr"],
                                                    b1.ctypes.data, C.byref(n1), b2.ctypes.data, C.byref(n2))
                if rc != -5:    # MIJ_ERR_OVERFLOW
                    break
                cap1 = max(cap1, n1.value + n1.value // 8 + 4096)
                cap2 = max(cap2, n2.value + n2.value // 8 + 4096, cap1 if n2.value == 0 else 0)
            _lib.check(rc, self._enc._h, "mij_secondary_encode_host_ex")
            return b1[:n1.value].tobytes(), b2[:n2.value].tobytes(), 1
        except MiJpegError as e:
            print("[ERROR] Exception caught: %s" % e)
            return b"", b"", 0

    def secondaryDecode(self, primary, secondary, gain=1):
        """(primary, secondary) -> (H x W x 3 uint8 BGR, run_state): clip(decode(primary) + (decode(secondary) - 128) / gain), with the
        gain the pair was written with."""
        try:
            if self._dec is None:
                raise MiJpegError("secondaryDecode() before buildDecodeEnv()")
            inf = Decoder.info(primary)
            out = np.empty((inf["height"], inf["width"], 3), np.uint8)
            p1, p2 = np.frombuffer(primary, np.uint8), np.frombuffer(secondary, np.uint8)
            w, h = C.c_int(), C.c_int()
            sp = _lib.SecondaryParams(0, -1, gain)
            self._dec._check(self._dec._L.mij_secondary_decode_host_ex(self._dec._h, C.byref(sp), p1.ctypes.data, len(primary), p2.ctypes.data,
                                                                       len(secondary), out.ctypes.data, inf["width"] * 3, _FMT["bgr"],
                                                                       C.byref(w), C.byref(h)), "mij_secondary_decode_host_ex")
            return out, 1
        except MiJpegError as e:
            print("[ERROR] Exception caught: %s" % e)
            return None, 0

    def save(self, save_path, obuffer):
        try:
            with open(save_path, "wb") as f:
                f.write(obuffer)
        except OSError as e:
            print("Exception caught: %s" % e)

    def __del__(self):
        try:
            self.deleteCompressEnv()
            self.deleteDecodeEnv()
        except Exception:
            pass
