"""Strip sharding of one JPEG across ranks: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no counterpart: it encodes the whole image in one nvjpegEncodeImage call on one GPU
(reference ImageCompressorImpl.cu:280). Here the image is cut into strips of whole MCU rows, every strip starting on a
restart-interval boundary, so that the only data exchanged between ranks is

  1. ONE all-reduce (sum) of the 4 x 257 uint32 symbol statistics  -- only with optimised Huffman tables, because a
     single-scan baseline file can carry only one set of tables;
  2. ONE all-gather of the strips' byte counts (8 bytes per rank), device tensor to device tensor;
  3. the gather of the finished strip bitstreams to ONE rank, the image's root.

Everything else is local arithmetic. Two orchestrations of those steps:

* `DevicePipeline` (the measured path): sizes and offsets never visit the host. The image's root assembles the file in its
  own output buffer; the other ranks map that buffer (hipIpc*) and PUT their strip into it at the offset a kernel computes
  from the all-gathered sizes, over their own xGMI link. The root ROTATES over the ranks from image to image: a root takes in
  (N - 1) / N of a file over its inbound links (one per peer, ~77 GB/s each way), which at 8 GPUs is 0.33 ms per 204-MB file --
  more than the 0.16 ms a rank computes on its eighth of the image -- so with a fixed root the links into rank 0 would set the
  pace; rotating spreads the inbound traffic over every GPU's links. DEPTH images are in flight per rank, each on its own
  stream, so that one image's collectives and its put run under the other images' kernels; all collectives go through ONE
  communicator in one global order (an image's all-gather is issued behind the next image's all-reduce, see the class).
  Nothing in a steady-state step waits on the host.
* `StripPipeline` / `encode_step` (fallback when the peer mapping is unavailable, and the simple one-image form): sizes
  come to the host, strips travel as RCCL send/recv.

This module only sequences those steps; the JPEG work is done by a "strip encoder" object (the HIP one below; the CPU
tests plug in the oracle) with

    transform(stream)          -> 1-D int32 tensor of 4*257 statistics (device resident for the HIP encoder)
    entropy(stream)            -> (header, scan) uint8 tensors                      [host-synchronised path]
    entropy_sizes(slot, stream)   writes the strip's byte count into the 1-element int64 tensor `slot`
    place(target, sizes, rank, world, stream)   strip -> the assembled file at offset sum(sizes[:rank]); target None = this
                                                rank is the image's root (the file is assembled in its own buffer)
    file(target, sizes, rank, world)            the root (target None): the assembled file (waits for this handle's work)
"""
from .encoder import Encoder, geometry_query, ipc_close, ipc_export, ipc_open

DEPTH = 4          # images in flight per rank in DevicePipeline (a put takes longer than a rank's share of the kernels)


def partition_mcu_rows(mcu_rows, world, rank, rows_per_unit=1):
    """MCU rows [r0, r1) of rank `rank`: contiguous, as even as possible, in units of `rows_per_unit` MCU rows (the number
    of MCU rows a restart interval spans when it is longer than one row; 1 when the interval divides the row). With fewer
    units than ranks the FIRST ranks get one unit each and the rest are empty (r0 == r1): rank 0, which assembles the
    file and writes its header, always owns the first strip."""
    units = (mcu_rows + rows_per_unit - 1) // rows_per_unit
    if units >= world:
        u0, u1 = rank * units // world, (rank + 1) * units // world
    else:
        u0, u1 = min(rank, units), min(rank + 1, units)
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

    def __init__(self, torch, enc, d_img, fmt="bgr", shared_statistics=True):
        self.torch, self.enc, self.d_img, self.fmt = torch, enc, d_img, fmt
        self.pitch = d_img.stride(0) * d_img.element_size()
        self.shared_statistics = shared_statistics
        self.d_hist = torch.zeros(4 * 257, dtype=torch.int32, device=d_img.device)
        # The statistics go to a tensor a collective can reduce in place. A single rank has nothing to reduce and leaves the
        # handle on its own buffers (shared_statistics=False): those alternate and need no clearing per image.
        if shared_statistics:
            enc.set_histogram_buffer(self.d_hist.data_ptr())

    def transform(self, stream=0):
        self.enc.transform(self.d_img.data_ptr(), self.pitch, self.fmt, 0, stream)
        return self.d_hist

    # ---- device-side protocol (DevicePipeline) ----
    def entropy_sizes(self, slot, stream=0):
        self.enc.entropy_sizes(slot.data_ptr(), stream)

    def place(self, target, sizes, rank, world, stream=0):
        ptr, cap = target if target is not None else (0, 0)       # None: this rank is the root and assembles in its own buffer
        self.enc.place(ptr, cap, sizes.data_ptr(), rank, world, stream)

    def file(self, target, sizes, rank, world):
        r = self.enc.sharded_result(sizes.data_ptr(), rank, world)
        self.last_result = r
        return self._view(r["d_buffer"] + r["header_offset"], r["file_bytes"], self.d_img.device) if target is None else None

    # ---- host-synchronised protocol (encode_step / StripPipeline) ----
    def entropy(self, stream=0):
        self.enc.entropy(stream)
        return self.collect_strip()

    def issue_entropy(self, stream=0):
        """Enqueue the entropy stage and return at once; `collect_strip` waits for it."""
        self.enc.entropy(stream)

    def collect_strip(self):
        r = self.enc.result()    # waits for this strip only (per-encode event); sizes are now known on the host
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


def _check_statistics(strips, world, optimize):
    """A strip encoder created with shared_statistics=False keeps its statistics in the handle's own buffers: what its
    transform() returns is NOT what the handle counts into, so reducing it over ranks would leave every rank on its local
    statistics -- different DHTs inside one file. Only a single rank (nothing to reduce) may run that way."""
    if world > 1 and optimize:
        for s in (strips or []):
            if s is not None and getattr(s, "shared_statistics", True) is False:
                raise ValueError("a strip encoder with shared_statistics=False cannot take part in a multi-rank encode with optimised tables")


def device_bytes(torch, ptr, nbytes, device):
    """uint8 tensor view over device memory owned by libmijpeg (valid until the next encode on that handle)."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


# =====================================================================================================================
# Device-side pipeline: no host round trip in a step
# =====================================================================================================================
class DevicePipeline:
    """DEPTH images in flight per rank; per image and rank, enqueued without a single host wait:

        slot k = i % DEPTH, on stream S_k:
          transform(i) -> all_reduce(statistics) -> entropy_sizes(i) -> all_gather(sizes) -> place(i)

    Image i is assembled on its ROOT, rank i % (number of ranks that own a strip) -- rank 0 always with rotate=False. `place`
    on the root compacts the strip straight to its place in the root's own buffer of slot k; on every other rank it compacts
    locally and puts the strip into that buffer at the offset a kernel derives from the gathered sizes. Every rank issues the
    same collectives in the same order.

    Communicators (`comms`):
      * "ordered" (default): ONE communicator for every slot, so the collectives of all images execute in one global order
        on every rank -- the order they are issued in, which is the same everywhere. So that an image's all-reduce does not
        queue behind the previous image's all-gather (which waits for that image's entropy coder), the all-gather and
        placement of image i are issued one step late, BEHIND the all-reduce of image i + 1:
            AR(0) | AR(1) AG(0) | AR(2) AG(1) | ...
        Each image's entropy chain then starts as soon as its own statistics are reduced; two images' chains overlap.
      * "per-slot": one communicator per slot, collectives of different images independent of each other. Several
        communicators in flight on several streams is a pattern PyTorch documents as unsafe without external ordering
        (ranks may run the collectives of different communicators in different orders); it is kept as an experiment switch
        (`bench.py --comms per-slot`) until a multi-GPU node has run it.

    Hazards, all resolved by stream order: a handle (coefficients, scratch, output buffer) is reused by image i + DEPTH on
    the same stream; a peer's put of image i + DEPTH into a root's buffer k can only start after the all-gather of that
    image, i.e. after that root has executed everything of image i on its S_k -- including, when it was not image i's root,
    its own put out of that buffer. The file of image i is complete on its root once a LATER collective of slot k has
    completed there (each peer enqueues it behind its put) -- `collect` issues one, and then agrees on a status word over all
    ranks, so that a strip some rank failed to place fails the image EVERYWHERE (the root cannot see a peer's flags).

    `strips` holds DEPTH strip encoders, or None on a rank that owns no strip (more ranks than restart-aligned strips):
    such a rank contributes zero statistics and a zero size and is never a root (empty ranks come last, partition_mcu_rows).
    `targets[k][r]` is what `place` / `file` take for slot k when rank r is the root: None on r itself
    (`open_file_targets` for the HIP encoder).
    """

    def __init__(self, torch, dist, strips, targets, optimize, device=None, use_streams=None, rotate=True, comms="ordered"):
        assert comms in ("ordered", "per-slot")
        self.torch, self.dist, self.strips, self.targets, self.optimize = torch, dist, strips, targets, optimize
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        _check_statistics(strips, self.world, optimize)
        self.depth = len(targets)
        self.comms = comms
        has = [None] * self.world
        dist.all_gather_object(has, strips is not None)
        self.nroots = sum(1 for h in has if h) if rotate else 1     # ranks 0 .. nroots-1 own strips
        self.roots = [0] * self.depth                              # root of the image in each slot
        self.last_root = 0
        if device is None:
            device = next((s.d_img.device for s in (strips or []) if s is not None and getattr(s, "d_img", None) is not None), torch.device("cpu"))
        self.device = device
        cuda = device.type == "cuda"
        self.use_streams = cuda if use_streams is None else use_streams
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.depth)] if self.use_streams else [None] * self.depth
        if self.use_streams:
            for s in self.streams:
                s.wait_stream(torch.cuda.current_stream(device))      # the image was produced on the current stream
        if comms == "per-slot":
            self.groups = [dist.new_group() for _ in range(self.depth)]   # same order on every rank
        else:
            self.groups = [None] * self.depth                             # the default group: one order for everything
        self.mine = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.depth)]
        self.sizes = [torch.zeros(self.world, dtype=torch.int64, device=device) for _ in range(self.depth)]
        self.zero_hist = [torch.zeros(4 * 257, dtype=torch.int32, device=device) for _ in range(self.depth)]
        self.done = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(self.depth)]
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.i = 0
        self.deferred = None      # "ordered": slot whose all-gather + placement are still to be issued
        self.pending = []         # slots whose image has been issued and not yet collected, oldest first

    def _ctx(self, k):
        import contextlib
        return self.torch.cuda.stream(self.streams[k]) if self.use_streams else contextlib.nullcontext()

    def _all_gather(self, out, inp, group):
        if hasattr(self.dist, "all_gather_into_tensor"):
            try:
                self.dist.all_gather_into_tensor(out, inp, group=group)
                return
            except NotImplementedError:      # a backend without the flat form; real failures of the collective propagate
                pass
        parts = [self.torch.empty_like(inp) for _ in range(self.world)]
        self.dist.all_gather(parts, inp, group=group)
        out.copy_(self.torch.cat(parts))

    def _gather_place(self, k):
        """Second half of slot k's image: sizes all-gathered device to device, strip placed / put."""
        st = self.strips[k] if self.strips else None
        sh = self.streams[k].cuda_stream if self.use_streams else 0
        with self._ctx(k):
            self._all_gather(self.sizes[k], self.mine[k], self.groups[k])
            if st is not None:
                st.place(self.targets[k][self.roots[k]], self.sizes[k], self.rank, self.world, sh)

    def _issue_deferred(self):
        if self.deferred is not None:
            k, self.deferred = self.deferred, None
            self._gather_place(k)

    def step(self):
        """Issue the next image on slot i % DEPTH and return at once (no host wait, no collective beyond the images' own
        two per step). An older image still uncollected in that slot is overwritten: its file stays valid only until the
        peers' puts of the new image start, so callers that want EVERY file call `collect` before the slot comes round again."""
        k = self.i % self.depth
        if self.deferred == k:           # depth 1: the slot's previous image must be placed before its buffers are reused
            self._issue_deferred()
        root = self.roots[k] = self.last_root = self.i % self.nroots
        self.pending = [q for q in self.pending if q != k]
        self.i += 1
        st = self.strips[k] if self.strips else None
        sh = self.streams[k].cuda_stream if self.use_streams else 0
        with self._ctx(k):
            hist = st.transform(sh) if st is not None else self.zero_hist[k].zero_()
            if self.optimize:
                self.dist.all_reduce(hist, group=self.groups[k])
            if st is not None:
                st.entropy_sizes(self.mine[k], sh)
            else:
                self.mine[k].zero_()
        if self.comms == "ordered":
            self._issue_deferred()       # the previous image's all-gather, BEHIND this image's all-reduce
            self.deferred = k
        else:
            self._gather_place(k)
        self.pending.append(k)
        return k

    def collect(self, want_file=True):
        """Complete the OLDEST pending image: one small collective behind every rank's put, a wait for this rank's stream,
        a status word agreed over all ranks, then (on the image's root, if wanted) the assembled file; None on every other
        rank. Every rank must call this the same number of times, in the same order relative to `step`. Not part of a
        steady-state step: the bench only flushes at the end of the timed region."""
        if not self.pending:
            return None
        k = self.pending.pop(0)
        if self.deferred == k:
            self._issue_deferred()
        with self._ctx(k):
            self.dist.all_reduce(self.done[k], group=self.groups[k])
        if self.use_streams:
            self.streams[k].synchronize()
        st = self.strips[k] if self.strips else None
        out, err = None, None
        if st is not None:
            try:
                out = st.file(self.targets[k][self.roots[k]], self.sizes[k], self.rank, self.world)   # also checks this handle's status
            except Exception as e:       # noqa: BLE001 -- reported on every rank below
                err = e
        # A peer whose put refused to write (or whose handle failed) knows it; the root does not. Agree before anyone
        # hands out a file with a hole in it.
        self.status.fill_(0 if err is None else 1)
        self.dist.all_reduce(self.status, op=self.dist.ReduceOp.MAX, group=self.groups[k])
        if int(self.status.item()):
            raise RuntimeError("sharded encode failed on %s: %s" % ("this rank" if err is not None else "another rank",
                                                                    err if err is not None else "see that rank's log"))
        return out if (want_file and self.rank == self.roots[k]) else None

    def flush(self):
        """Complete everything in flight; returns (on rank `last_root`) the file of the LAST image issued."""
        out = None
        self._issue_deferred()
        while self.pending:
            out = self.collect(want_file=len(self.pending) == 1)
        if self.use_streams:
            for s in self.streams:
                self.torch.cuda.current_stream(self.device).wait_stream(s)
        return out


def full_scan_capacity(geometry):
    """Bytes to reserve for the entropy-coded data of the WHOLE image (what one handle reserves for its strip, scaled up):
    two bytes per coefficient cover every photographic setting; noise at q100 can exceed it (MIJ_ERR_OVERFLOW then)."""
    return geometry["mcus_per_row"] * geometry["mcu_rows"] * geometry["blocks_per_mcu"] * 64 + 65536


def open_file_targets(torch, dist, strips, rank, world, device_index, whole_geometry, depth=None):
    """Peer mapping of the output buffers of every rank that owns a strip -- any of them can be an image's root -- on every
    other rank (mij_ipc_export / mij_ipc_open). Returns targets[k][r] = (pointer, capacity) of rank r's scan area of slot k as
    this rank sees it (None for r = this rank, and for ranks without a strip), or None when ANY rank could not map (then every
    rank takes the send/recv fallback). Every strip-owning rank first reserves room for the whole file in each of its handles."""
    cap = full_scan_capacity(whole_geometry)
    mine, ok = None, True
    if strips is not None:
        try:
            mine = []
            for st in strips:
                st.enc.reserve_output(cap)
                base, off, c = st.enc.output_buffer()
                mine.append((ipc_export(base), off, c))
        except Exception as e:       # noqa: BLE001 -- any failure means "no peer mapping": all ranks fall back together
            mine, ok = ("error", str(e)), False
    every = [None] * world
    dist.all_gather_object(every, mine)
    if any(isinstance(h, tuple) and h and h[0] == "error" for h in every):
        ok = False
    nslots = depth if depth is not None else max((len(h) for h in every if isinstance(h, list)), default=0)
    targets = [[None] * world for _ in range(nslots)]
    opened = []
    if ok:
        try:
            for r, handles in enumerate(every):
                if r == rank or not isinstance(handles, list):
                    continue
                for k, (h, off, c) in enumerate(handles):
                    p = ipc_open(device_index, h)
                    opened.append(p)
                    targets[k][r] = (p + off, c)
        except Exception:            # noqa: BLE001
            ok = False
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok))
    if not all(oks):
        for p in opened:
            try:
                ipc_close(p)
            except Exception:        # noqa: BLE001
                pass
        return None
    return targets


# =====================================================================================================================
# Host-synchronised forms (fallback / one image at a time)
# =====================================================================================================================
def encode_step(torch, dist, strip_encoder, optimize, out_cache, stream=0):
    """One whole-image encode across all ranks of the default process group, one image at a time. Returns, on rank 0, a
    uint8 tensor holding the complete JFIF file (a view into `out_cache["buf"]`, reused across calls); None on the other
    ranks. `strip_encoder` may be None on a rank that owns no strip."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    _check_statistics([strip_encoder], world, optimize)
    hist = strip_encoder.transform(stream) if strip_encoder is not None else torch.zeros(4 * 257, dtype=torch.int32, device=out_cache.get("device", "cpu"))
    if world > 1 and optimize:
        dist.all_reduce(hist)                      # the only collective on the data path before entropy coding
    if world == 1 and hasattr(strip_encoder, "encode_whole"):
        return strip_encoder.encode_whole(stream)
    if strip_encoder is not None:
        header, scan = strip_encoder.entropy(stream)
    else:
        header = scan = torch.zeros(0, dtype=torch.uint8, device=hist.device)
    if world == 1:
        n = header.numel() + scan.numel()
        buf = _buffer(torch, out_cache, n, scan.device)
        buf[:header.numel()].copy_(header)
        buf[header.numel():n].copy_(scan)
        return buf[:n]
    return _gather_to_rank0(torch, dist, header, scan, out_cache)


class StripPipeline:
    """Host-synchronised multi-rank encode with TWO images in flight per rank (the fallback when rank 0's buffers cannot be
    peer-mapped): the gather of image i-1 to rank 0 as RCCL send/recv and the host round trips for its sizes run while
    image i's kernels do. Every rank executes the same sequence of collectives, in the same order:

        step i:   transform(i) -> all_reduce(statistics i) -> entropy(i)           [enqueued, no host wait]
                  sizes(i-1): all_gather -> host;  gather(i-1): point-to-point      [on a side stream]

    Two strip encoders (handles) alternate; a handle is reused only after the sends that read its bitstream are complete
    (the compute stream waits for the side stream at the start of a step). `step` returns, on rank 0, the complete file
    of the PREVIOUS image (None for the first call); `flush` returns the last one.
    """

    def __init__(self, torch, dist, strips, optimize):
        assert len(strips) == 2
        _check_statistics(strips, dist.get_world_size() if dist.is_initialized() else 1, optimize)
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
    """all_gather of the strip sizes, then the strips travel to rank 0 point-to-point."""
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


# =====================================================================================================================
# Progressive output, sharded (round 5): the reference's own encoding (ImageCompressorImpl.cu:28) over N ranks
# =====================================================================================================================
def progressive_strip_interval(mcus_per_row):
    """Restart interval of a SHARDED progressive encode: it must divide the MCUs per row (every one of the ten scans then cuts into
    restart intervals at strip boundaries; mi_jpeg.h). The largest divisor up to 1024 MCUs: long intervals suit the progressive coder
    (520 at the headline width, where one rank alone picks 640)."""
    best = 1
    for d in range(1, min(mcus_per_row, 1024) + 1):
        if mcus_per_row % d == 0:
            best = d
    return best


def progressive_offsets(sizes, header_bytes):
    """sizes[r][i]: bytes of rank r's segment of scan i (mij_encode_prog_emit, all-gathered); header_bytes[i]: bytes in front of scan i's
    data. -> (offsets[r][i], file_bytes): where every segment goes in the file, scans in order, within a scan the ranks in order."""
    world = len(sizes)
    offs = [[0] * 10 for _ in range(world)]
    off = 0
    for i in range(10):
        off += header_bytes[i]
        for r in range(world):
            offs[r][i] = off
            off += sizes[r][i]
    return offs, off + 2          # + EOI


class HipProgressiveStrip:
    """This rank's strip of a sharded PROGRESSIVE encode, backed by libmijpeg (HIP)."""

    def __init__(self, torch, enc, d_img, fmt="bgr"):
        self.torch, self.enc, self.d_img, self.fmt = torch, enc, d_img, fmt
        self.pitch = d_img.stride(0) * d_img.element_size()
        p, n = enc.prog_histogram_buffer()
        self.hist = device_words(torch, p, n, d_img.device)        # int32 view over the handle's buffer: reduced in place
        self.stage = None

    def statistics(self, stream=0):
        self.enc.transform(self.d_img.data_ptr(), self.pitch, self.fmt, 0, stream)
        self.enc.prog_statistics(stream)
        return self.hist

    def emit(self, stream=0):
        return self.enc.prog_emit(stream)

    def place_into(self, file_tensor, offsets, file_bytes, flags, stream=0):
        self.enc.prog_place(offsets, file_tensor.data_ptr(), file_tensor.numel(), file_bytes, flags, stream)

    def place_staged(self, sizes, stream=0):
        """The ten segments into a staging tensor of this rank, 64 bytes apart (a segment's dropped last marker may spill two bytes);
        -> the ten views to send."""
        torch = self.torch
        local, off = [], 0
        for n in sizes:
            local.append(off)
            off += n + 64
        if self.stage is None or self.stage.numel() < off + 64:
            self.stage = torch.empty(off + (off >> 3) + 4096, dtype=torch.uint8, device=self.d_img.device)
        self.enc.prog_place(local, self.stage.data_ptr(), self.stage.numel(), 0, 0, stream)
        return [self.stage[o:o + n] for o, n in zip(local, sizes)]


def device_words(torch, ptr, nwords, device):
    """int32 tensor view over device memory owned by libmijpeg."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nwords),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


def encode_step_progressive(torch, dist, strip, cache, stream=0):
    """One whole-image PROGRESSIVE encode across all ranks of the default process group, one image at a time, host-synchronised:
    statistics -> ONE all-reduce (10 x 4 x 257 words) -> tables + this strip's intervals of all ten scans -> all-gather of the 10 sizes per rank ->
    every rank's ten segments travel to rank 0 (point-to-point), which writes headers and EOI. Returns, on rank 0, a uint8 tensor with
    the complete file (a view into cache["buf"]); None elsewhere. `strip` may be None on a rank that owns no strip. The strip object
    may be any implementation of HipProgressiveStrip's four methods (the CPU tests play it with the oracle)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    dev = cache.get("device", "cpu")
    hist = strip.statistics(stream) if strip is not None else torch.zeros(10 * 4 * 257, dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(hist)                                  # the only collective in front of the entropy stage
    sizes, hdr = strip.emit(stream) if strip is not None else ([0] * 10, [0] * 10)
    if world == 1:
        offs, total = progressive_offsets([sizes], hdr)
        buf = _buffer(torch, cache, total + 64, dev)
        strip.place_into(buf, offs[0], total, 3, stream)       # headers + EOI
        return buf[:total]
    mine = torch.tensor(sizes + hdr, dtype=torch.int64, device=dev)
    every = torch.zeros(world * 20, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(every, mine)
    ev = every.cpu().view(world, 20).tolist()
    all_sizes = [[int(v) for v in row[:10]] for row in ev]
    owners = [r for r in range(world) if sum(all_sizes[r]) > 0]
    if not owners:
        raise RuntimeError("no rank owns a strip")
    first = owners[0]
    hdr = [int(v) for v in ev[first][10:]]                     # (identical on every rank that owns a strip)
    offs, total = progressive_offsets(all_sizes, hdr)
    if first != 0:              # (partition_mcu_rows hands the strips out from rank 0 on: the same arithmetic on every rank)
        raise RuntimeError("rank 0 must own the first strip")
    if rank == 0:
        buf = _buffer(torch, cache, total + 64, dev)
        strip.place_into(buf, offs[0], total, 3, stream)       # its own ten segments, the ten headers, the EOI
        ops = []
        for r in range(1, world):
            for i in range(10):
                if all_sizes[r][i]:
                    ops.append(dist.P2POp(dist.irecv, buf[offs[r][i]:offs[r][i] + all_sizes[r][i]], r))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return buf[:total]
    if strip is not None:
        segs = strip.place_staged(sizes, stream)
        ops = [dist.P2POp(dist.isend, sg, 0) for sg in segs if sg.numel()]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    return None


def strip_rows(width, height, quality, optimize, css, rank, world, restart_interval=-1):
    """(whole-image geometry, first MCU row, one-past-last MCU row) of `rank`: pure arithmetic (mij_geometry_query), no
    device and no communication. r0 == r1: this rank owns no strip."""
    g0 = geometry_query(width, height, quality, optimize, css, restart_interval)
    unit = rows_per_restart_unit(g0["mcus_per_row"], g0["restart_interval"])
    r0, r1 = partition_mcu_rows(g0["mcu_rows"], world, rank, unit)
    return g0, r0, r1


def make_hip_strip_encoder(torch, width, height, quality, optimize, css, rank, world, device_index, fmt="bgr",
                           restart_interval=-1, progressive=False):
    """This rank's Encoder for its strip (its `geometry` tells which pixel rows to fill), or None if the rank owns no strip
    (more ranks than restart-aligned strips)."""
    if progressive and world == 1:
        return Encoder(width, height, quality, optimize, css, restart_interval, device_index, progressive=True)   # (its own AUTO rule: a multiple of 64 blocks)
    if progressive:
        # sharded: the interval must divide the MCU row (progressive_strip_interval), strips are whole MCU rows
        g00 = geometry_query(width, height, quality, optimize, css, restart_interval)
        ri = restart_interval if restart_interval > 0 else progressive_strip_interval(g00["mcus_per_row"])
        g0, r0, r1 = strip_rows(width, height, quality, optimize, css, rank, world, ri)
        if r1 <= r0:
            return None
        return Encoder(width, height, quality, optimize, css, ri, device_index, r0, r1 - r0, progressive=True)
    g0, r0, r1 = strip_rows(width, height, quality, optimize, css, rank, world, restart_interval)
    if r1 <= r0:
        return None
    return Encoder(width, height, quality, optimize, css, g0["restart_interval"], device_index, r0, r1 - r0)
