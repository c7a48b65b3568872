"""Strip sharding of one JPEG across ranks: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no counterpart: it encodes the whole image in one nvjpegEncodeImage call on one GPU
(reference ImageCompressorImpl.cu:280). Here the image is cut into strips of whole MCU rows, every strip starting on a
restart-interval boundary, so that the only data exchanged between ranks is

  1. ONE all-reduce (sum) of the 4 x 257 uint32 symbol statistics  -- only with optimised Huffman tables, because a
     single-scan baseline file can carry only one set of tables;
  2. ONE all-gather of the strips' byte counts (8 bytes per rank);
  3. the gather of the finished strip bitstreams to rank 0 (point-to-point sends: xGMI links run in parallel).

Everything else is local arithmetic. This module only sequences those steps; the JPEG work is done by a "strip
encoder" object (the HIP one below; the CPU tests plug in the oracle) with

    transform(stream)  -> 1-D int32 tensor of 4*257 statistics (device resident for the HIP encoder)
    entropy(stream)    -> (header, scan): 1-D uint8 tensors; `header` is the SOI..SOS prefix (same on every rank),
                          `scan` this strip's entropy-coded bytes including its trailing RSTn or, last strip, EOI.
"""
import ctypes as C

from .encoder import Encoder


def partition_mcu_rows(mcu_rows, world, rank, rows_per_unit=1):
    """MCU rows [r0, r1) of rank `rank`: contiguous, as even as possible, in units of `rows_per_unit` MCU rows (the number
    of MCU rows a restart interval spans when it is longer than one row; 1 when the interval divides the row)."""
    units = (mcu_rows + rows_per_unit - 1) // rows_per_unit
    u0, u1 = rank * units // world, (rank + 1) * units // world
    return min(u0 * rows_per_unit, mcu_rows), min(u1 * rows_per_unit, mcu_rows)


def rows_per_restart_unit(mcus_per_row, restart_interval):
    """1 if the restart interval divides the MCUs per row (what MIJ_RESTART_AUTO guarantees), else the number of MCU rows
    after which interval boundaries and row boundaries coincide again."""
    if mcus_per_row % restart_interval == 0:
        return 1
    from math import gcd
    return restart_interval // gcd(restart_interval, mcus_per_row)


class HipStripEncoder:
    """Strip encoder backed by libmijpeg (HIP). `d_img` is this rank's strip of the image, device resident."""

    def __init__(self, torch, enc, d_img, fmt="bgr"):
        self.torch, self.enc, self.d_img, self.fmt = torch, enc, d_img, fmt
        self.pitch = d_img.stride(0) * d_img.element_size()
        self.d_hist = torch.zeros(4 * 257, dtype=torch.int32, device=d_img.device)
        enc.set_histogram_buffer(self.d_hist.data_ptr())

    def transform(self, stream=0):
        self.enc.transform(self.d_img.data_ptr(), self.pitch, self.fmt, 0, stream)
        return self.d_hist

    def entropy(self, stream=0):
        self.enc.entropy(stream)
        r = self.enc.result()    # waits for this strip; sizes are now known on the host
        self.last_result = r
        dev = self.d_img.device
        return (self._view(r["d_buffer"] + r["header_offset"], r["header_bytes"], dev),
                self._view(r["d_buffer"] + r["scan_offset"], r["scan_bytes"], dev))

    def issue_entropy(self, stream=0):
        """Enqueue the entropy stage and return at once (pipelined multi-rank step); `collect_strip` waits for it."""
        self.enc.entropy(stream)

    def collect_strip(self):
        r = self.enc.result()    # waits for this strip only (per-encode event)
        self.last_result = r
        dev = self.d_img.device
        return (self._view(r["d_buffer"] + r["header_offset"], r["header_bytes"], dev),
                self._view(r["d_buffer"] + r["scan_offset"], r["scan_bytes"], dev))

    def encode_whole(self, stream=0):
        """Single-rank shortcut: entropy-code and return the complete file (header and scan sit back to back in the
        encoder's buffer), without building the separate header / scan views nobody would read."""
        self.enc.entropy(stream)
        r = self.enc.result()
        self.last_result = r
        return self._view(r["d_buffer"] + r["header_offset"], r["file_bytes"], self.d_img.device)

    def issue_whole(self, stream=0):
        """Single-rank, two handles in flight: enqueue a complete encode (all kernels + the copy of the result record) and
        return at once; `finish_whole` later waits for THIS encode only (mij_encode_result waits on a per-encode event)."""
        self.enc.transform(self.d_img.data_ptr(), self.pitch, self.fmt, 0, stream)
        self.enc.entropy(stream)

    def finish_whole(self):
        r = self.enc.result()
        self.last_result = r
        return self._view(r["d_buffer"] + r["header_offset"], r["file_bytes"], self.d_img.device)

    def whole_file(self):
        r = self.last_result
        return self._view(r["d_buffer"] + r["header_offset"], r["file_bytes"], self.d_img.device)

    def _view(self, ptr, nbytes, dev):
        # wrapping device memory in a tensor costs ~10 us of host time; the same (address, size) recurs on every step of a
        # steady-state loop, so keep the last few views
        cache = self.__dict__.setdefault("_views", {})
        key = (int(ptr), int(nbytes))
        t = cache.get(key)
        if t is None:
            if len(cache) > 8:
                cache.clear()
            t = cache[key] = device_bytes(self.torch, ptr, nbytes, dev)
        return t


def device_bytes(torch, ptr, nbytes, device):
    """uint8 tensor view over device memory owned by libmijpeg (valid until the next encode on that handle)."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


def encode_step(torch, dist, strip_encoder, optimize, out_cache, stream=0):
    """One whole-image encode across all ranks of the default process group. Returns, on rank 0, a uint8 tensor holding the
    complete JFIF file (a view into `out_cache["buf"]`, reused across calls); None on the other ranks."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    hist = strip_encoder.transform(stream)
    if world > 1 and optimize:
        dist.all_reduce(hist)                      # the only collective on the data path before entropy coding
    if world == 1 and hasattr(strip_encoder, "encode_whole"):
        return strip_encoder.encode_whole(stream)
    header, scan = strip_encoder.entropy(stream)
    if world == 1:
        n = header.numel() + scan.numel()
        buf = _buffer(torch, out_cache, n, scan.device)
        buf[:header.numel()].copy_(header)
        buf[header.numel():n].copy_(scan)
        return buf[:n]
    return _gather_to_rank0(torch, dist, header, scan, out_cache)


class StripPipeline:
    """Multi-rank encode with TWO images in flight per rank, so that the gather of image i-1 to rank 0 (the longest leg:
    (N-1)/N of the file crosses rank 0's xGMI links) and the host round trips for its sizes run while image i's kernels
    do. Every rank executes the same sequence of collectives, in the same order:

        step i:   transform(i) -> all_reduce(statistics i) -> entropy(i)           [enqueued, no host wait]
                  sizes(i-1): all_gather -> host;  gather(i-1): point-to-point      [on a side stream]

    Two strip encoders (handles) alternate; a handle is reused only after the sends that read its bitstream are complete
    (the compute stream waits for the side stream at the start of a step). `step` returns, on rank 0, the complete file
    of the PREVIOUS image (None for the first call); `flush` returns the last one.
    """

    def __init__(self, torch, dist, strips, optimize):
        assert len(strips) == 2
        self.torch, self.dist, self.strips, self.optimize = torch, dist, strips, optimize
        self.caches = [{}, {}]
        self.i, self.pending = 0, None
        d_img = getattr(strips[0], "d_img", None)
        self.side = torch.cuda.Stream() if (d_img is not None and d_img.is_cuda) else None   # CPU tests: no streams

    def step(self):
        torch, dist = self.torch, self.dist
        k = self.i & 1
        self.i += 1
        cur = self.strips[k]
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)     # the sends of image i-2 (same handle) are done
        stream = torch.cuda.current_stream().cuda_stream if self.side is not None else 0
        hist = cur.transform(stream)
        if self.optimize:
            dist.all_reduce(hist)
        cur.issue_entropy(stream)
        out = self._finish()
        self.pending = (cur, self.caches[k])
        return out

    def flush(self):
        out = self._finish()
        if self.side is not None:
            self.torch.cuda.current_stream().wait_stream(self.side)
        return out

    def _finish(self):
        if self.pending is None:
            return None
        (s, cache), self.pending = self.pending, None
        header, scan = s.collect_strip()
        if self.side is None:
            return _gather_to_rank0(self.torch, self.dist, header, scan, cache)
        with self.torch.cuda.stream(self.side):
            return _gather_to_rank0(self.torch, self.dist, header, scan, cache)


def _gather_to_rank0(torch, dist, header, scan, cache):
    """all_gather of the strip sizes, then the strips travel to rank 0 point-to-point (same protocol as encode_step)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([scan.numel()], dtype=torch.int64, device=scan.device)
    sizes = torch.zeros(world, dtype=torch.int64, device=scan.device)
    dist.all_gather_into_tensor(sizes, mine)
    sz = [int(v) for v in sizes.cpu().tolist()]
    if rank == 0:
        total = header.numel() + sum(sz)
        buf = _buffer(torch, cache, total, scan.device)
        off = header.numel()
        buf[:off].copy_(header)
        buf[off:off + sz[0]].copy_(scan)
        off += sz[0]
        ops = []
        for r in range(1, world):
            if sz[r]:
                ops.append(dist.P2POp(dist.irecv, buf[off:off + sz[r]], r))
            off += sz[r]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return buf[:total]
    if scan.numel():
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, scan.contiguous(), 0)]):
            w.wait()
    return None


def _buffer(torch, cache, n, device):
    buf = cache.get("buf")
    if buf is None or buf.numel() < n or buf.device != device:
        buf = torch.empty(n + (n >> 3) + 4096, dtype=torch.uint8, device=device)
        cache["buf"] = buf
    return buf


def make_hip_strip_encoder(torch, width, height, quality, optimize, css, rank, world, device_index, fmt="bgr",
                           restart_interval=-1, progressive=False):
    """Creates this rank's Encoder for its strip plus the geometry needed to fill the strip with pixels."""
    probe = Encoder(width, height, quality, optimize, css, restart_interval, device_index, 0, 1)
    g0 = probe.geometry
    probe.close()
    unit = rows_per_restart_unit(g0["mcus_per_row"], g0["restart_interval"])
    r0, r1 = partition_mcu_rows(g0["mcu_rows"], world, rank, unit)
    if r1 <= r0:
        raise ValueError("more ranks than restart-aligned strips: rank %d would be empty" % rank)
    if progressive:
        if world != 1:
            raise ValueError("progressive output is not sharded: every scan spans the whole image")
        return Encoder(width, height, quality, optimize, css, g0["restart_interval"], device_index, progressive=True)
    enc = Encoder(width, height, quality, optimize, css, g0["restart_interval"], device_index, r0, r1 - r0)
    return enc
