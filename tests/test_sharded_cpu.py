"""world_size 2 / 3 / 5 / 8 gloo runs of the strip-sharding orchestration (nvjpeg_imagecompressor_amd/sharded.py) on CPU.
The HIP strip encoder needs a GPU, so the per-strip JPEG work is done here by the oracle (test infrastructure); what
is under test is the product's partitioning (incl. ranks that own no strip), the statistics all-reduce, the size
all-gather and the gather of the strips -- both the device-side pipeline (DevicePipeline: DEPTH images in flight, strips
PUT into rank 0's buffer at offsets derived from the gathered sizes; peer-mapped memory is played by shared-memory
tensors) and the host-synchronised send/recv fallback: the N-rank file must equal the 1-rank file byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleStripEncoder:
    def __init__(self, O, img_strip, W, H, quality, css, optimize, ri, r0, r1, geo):
        self.O, self.img, self.W, self.H = O, img_strip, W, H
        self.q, self.css, self.opt, self.ri, self.r0, self.r1, self.geo = quality, css, optimize, ri, r0, r1, geo

    def transform(self, stream=0):
        self.coef = self.O.coefficients(self.img, self.q, self.css)
        h = self.O.histogram(self.coef, self.W, self.img.shape[0], self.css, self.ri)
        self.hist = torch.from_numpy(h.astype(np.int64).reshape(-1).astype(np.int32))
        return self.hist

    def entropy(self, stream=0):
        first, last = self.r0 == 0, self.r1 == self.geo["mcuy"]
        hist = self.hist.numpy().astype(np.uint32).reshape(4, 257)
        data = self.O.encode_strip(self.coef, self.W, self.img.shape[0], self.H, self.q, self.css, self.opt, self.ri, hist,
                                   self.r0 * self.geo["mcux"] // self.ri, first, last)
        # every rank can produce the header (it depends on the all-reduced statistics only); strip it off here
        hdr_len = 0
        if first:
            i = data.index(b"\xff\xda")
            hdr_len = i + 2 + ((data[i + 2] << 8) | data[i + 3])
        self.header = data[:hdr_len]
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        return buf[:hdr_len], buf[hdr_len:]


    # the pipelined step splits entropy() into "enqueue" and "collect"; on CPU the work simply happens at collect time
    def issue_entropy(self, stream=0):
        pass

    def collect_strip(self):
        return self.entropy()

    # ---- device-side protocol (sharded.DevicePipeline); `target` = the root's buffer (shared-memory tensors play the peer
    # mapping), None on the root itself, which then uses `self.own`. The oracle writes a header only with the first strip, so
    # here the rank that owns it drops the header into the root's buffer (the HIP encoder builds it on every rank).
    def entropy_sizes(self, slot, stream=0):
        self._hdr, self._scan = self.entropy()
        slot[0] = self._scan.numel()

    def place(self, target, sizes, rank, world, stream=0):
        buf = self.own if target is None else target
        off = int(sizes[:rank].sum())
        assert int(sizes[rank]) == self._scan.numel()
        buf["scan"][off:off + self._scan.numel()] = self._scan
        if self.r0 == 0:
            buf["hdr"][:self._hdr.numel()] = self._hdr
            buf["hl"][0] = self._hdr.numel()

    def file(self, target, sizes, rank, world):
        if target is not None:
            return None
        return torch.cat([self.own["hdr"][:int(self.own["hl"][0])], self.own["scan"][:int(sizes.sum())]])


def _pipeline_worker(rank, world, port, W, H, css, optimize, ri, q, nimg, out_path):
    """StripPipeline: `nimg` different images through two alternating strip encoders; rank 0 saves every file."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from nvjpeg_imagecompressor_amd import sharded
    geo = O.geometry(W, H, css)
    unit = sharded.rows_per_restart_unit(geo["mcux"], ri)
    r0, r1 = sharded.partition_mcu_rows(geo["mcuy"], world, rank, unit)
    mcu_h = 8 * geo["vs"]
    y0, y1 = r0 * mcu_h, min(r1 * mcu_h, H)
    encs = [OracleStripEncoder(O, None, W, H, q, css, optimize, ri, r0, r1, geo) for _ in range(2)]
    pipe = sharded.StripPipeline(torch, dist, encs, optimize)
    outs = []
    for i in range(nimg):
        encs[i & 1].img = np.roll(O.synth_rgb(W, H), 7 * i, axis=1)[y0:y1].copy()    # image i
        o = pipe.step()
        outs.append(None if o is None else o.clone())     # the returned view is valid until its buffer's next image
    o = pipe.flush()
    outs.append(None if o is None else o.clone())
    assert outs[0] is None
    if rank == 0:
        for i, o in enumerate(outs[1:]):
            open(out_path + ".%d" % i, "wb").write(o.numpy().tobytes())
    else:
        assert all(o is None for o in outs)
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, W, H, css, optimize, ri, q, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from nvjpeg_imagecompressor_amd import sharded
    geo = O.geometry(W, H, css)
    unit = sharded.rows_per_restart_unit(geo["mcux"], ri)
    r0, r1 = sharded.partition_mcu_rows(geo["mcuy"], world, rank, unit)
    mcu_h = 8 * geo["vs"]
    y0, y1 = r0 * mcu_h, min(r1 * mcu_h, H)
    enc = None
    if r1 > r0:      # a rank may own no strip (fewer restart-aligned strips than ranks): it still joins every collective
        enc = OracleStripEncoder(O, O.synth_rgb(W, H, y0, y1 - y0), W, H, q, css, optimize, ri, r0, r1, geo)
    out = sharded.encode_step(torch, dist, enc, optimize, {})
    if rank == 0:
        open(out_path, "wb").write(out.numpy().tobytes())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _device_pipeline_worker(rank, world, port, W, H, css, optimize, ri, q, nimg, collect_each, targets, out_path, rotate, comms="ordered"):
    """DevicePipeline: `nimg` different images; either every file is collected before its slot comes round again
    (collect_each) or the loop runs like the bench -- issue only, one flush at the end, which yields the LAST file."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from nvjpeg_imagecompressor_amd import sharded
    geo = O.geometry(W, H, css)
    unit = sharded.rows_per_restart_unit(geo["mcux"], ri)
    r0, r1 = sharded.partition_mcu_rows(geo["mcuy"], world, rank, unit)
    mcu_h = 8 * geo["vs"]
    y0, y1 = r0 * mcu_h, min(r1 * mcu_h, H)
    depth = len(targets)
    encs = [OracleStripEncoder(O, None, W, H, q, css, optimize, ri, r0, r1, geo) for _ in range(depth)] if r1 > r0 else None
    assert rank != 0 or encs is not None              # rank 0 always owns the first strip
    if encs is not None:
        for k, e in enumerate(encs):
            e.own = targets[k][rank]
    mine = [[None if r == rank else t for r, t in enumerate(row)] for row in targets]      # what open_file_targets returns
    pipe = sharded.DevicePipeline(torch, dist, encs, mine, optimize, device=torch.device("cpu"), rotate=rotate, comms=comms)
    outs = []
    for i in range(nimg):
        if encs is not None:
            encs[i % depth].img = np.roll(O.synth_rgb(W, H), 7 * i, axis=1)[y0:y1].copy()    # image i
        pipe.step()
        if collect_each:
            o = pipe.collect()
            outs.append(None if o is None else o.clone())
    last = pipe.flush()
    if not collect_each:
        outs = [None] * (nimg - 1) + [None if last is None else last.clone()]
    for i, o in enumerate(outs):          # a file comes out on its image's root: rank i % (ranks with a strip), or rank 0
        if o is not None:
            assert rotate or rank == 0
            open(out_path + ".%d" % i, "wb").write(o.numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()


def _shared_targets(depth, world, nbytes):
    """targets[k][r]: rank r's buffer of slot k -- scan area, header area, header length -- visible to every process"""
    return [[{"scan": torch.zeros(nbytes, dtype=torch.uint8).share_memory_(), "hdr": torch.zeros(4096, dtype=torch.uint8).share_memory_(),
              "hl": torch.zeros(1, dtype=torch.int64).share_memory_()} for _ in range(world)] for _ in range(depth)]


@pytest.mark.parametrize("comms", ["ordered", "per-slot"])
@pytest.mark.parametrize("world,css,optimize,ri,collect_each,rotate", [
    (2, 1, True, 13, True, True), (3, 1, True, 13, True, True), (5, 2, True, 26, True, True), (8, 1, True, 13, True, True), (8, 0, False, 13, True, True),
    (2, 1, True, 40, True, True),        # one restart-aligned strip only: rank 1 owns nothing (the case round 1 dropped)
    (8, 1, True, 26, True, True),        # interval spans 2 MCU rows: 16 units over 8 ranks
    (8, 2, True, 104, True, True),       # 13 MCUs per row, 16 rows, interval 104 = 8 rows: 2 units, six empty ranks: two roots
    (3, 1, True, 13, False, True), (8, 1, True, 13, False, True),
    (3, 1, True, 13, True, False), (8, 1, True, 13, False, False),      # the fixed root (rank 0 assembles every file)
])
def test_device_pipeline_equals_one_rank_file(oracle, tmp_path, world, css, optimize, ri, collect_each, rotate, comms):
    """DEPTH images in flight, sizes all-gathered device-to-device, strips put at offsets derived from them, the assembling
    rank rotating from image to image: every image must come out as the 1-rank file (collect_each), and a bench-style loop
    (issue only, flush once) must end on the last one."""
    from nvjpeg_imagecompressor_amd import sharded
    W, H, q, nimg = 208, 250, 92, 9
    out = str(tmp_path / "dev.jpg")
    targets = _shared_targets(sharded.DEPTH, world, 3 * W * (H + 16))
    if comms == "per-slot" and (world not in (3, 8) or not rotate):
        pytest.skip("one communicator per slot (the experiment switch) is covered at world 3 and 8")
    mp.spawn(_device_pipeline_worker, args=(world, _free_port(), W, H, css, optimize, ri, q, nimg, collect_each, targets, out, rotate, comms),
             nprocs=world, join=True)
    for i in (range(nimg) if collect_each else [nimg - 1]):
        want = oracle.encode(np.roll(oracle.synth_rgb(W, H), 7 * i, axis=1).copy(), q, css, optimize, ri)
        assert open(out + ".%d" % i, "rb").read() == want, i


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("css,optimize,ri", [(1, True, 13), (2, True, 26), (0, False, 13), (1, True, 26),   # (1, 26): interval spans 2 MCU rows
                                             (1, True, 40)])                                                # one strip only: empty ranks
def test_n_rank_file_equals_one_rank_file(oracle, tmp_path, world, css, optimize, ri):
    W, H, q = 208, 250, 92          # H is not a multiple of the MCU height: the last strip owns the bottom edge
    out = tmp_path / "sharded.jpg"
    mp.spawn(_worker, args=(world, _free_port(), W, H, css, optimize, ri, q, str(out)), nprocs=world, join=True)
    want = oracle.encode(oracle.synth_rgb(W, H), q, css, optimize, ri)
    got = out.read_bytes()
    assert got == want


@pytest.mark.parametrize("world,optimize", [(2, True), (3, True), (2, False)])
def test_pipelined_step_two_images_in_flight(oracle, tmp_path, world, optimize):
    """The bench's N > 1 step keeps two images in flight per rank (StripPipeline): every image must still come out as
    the 1-rank file, in order, and the collectives must line up across ranks (a mismatch would hang or corrupt)."""
    W, H, q, css, ri, nimg = 208, 250, 92, 1, 13, 5
    out = str(tmp_path / "pipe.jpg")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), W, H, css, optimize, ri, q, nimg, out), nprocs=world, join=True)
    for i in range(nimg):
        want = oracle.encode(np.roll(oracle.synth_rgb(W, H), 7 * i, axis=1).copy(), q, css, optimize, ri)
        assert open(out + ".%d" % i, "rb").read() == want, i


def test_partition_arithmetic():
    from nvjpeg_imagecompressor_amd import sharded
    for rows in (1, 7, 625, 2500, 5000):
        for world in (1, 2, 3, 4, 8):
            for unit in (1, 2, 5):
                cuts = [sharded.partition_mcu_rows(rows, world, r, unit) for r in range(world)]
                assert cuts[0][0] == 0 and cuts[-1][1] == rows
                assert cuts[0][1] > 0                      # rank 0 (header, file assembly) always owns the first strip
                for (a0, a1), (b0, b1) in zip(cuts[:-1], cuts[1:]):
                    assert a1 == b0 and a0 <= a1
                assert all(c[0] % unit == 0 for c in cuts if c[1] > c[0])
                empty = [c[1] == c[0] for c in cuts]
                assert empty == sorted(empty)              # empty ranks, if any, come last
    assert sharded.rows_per_restart_unit(520, 104) == 1
    assert sharded.rows_per_restart_unit(520, 64) == 8      # the headline configuration: 625 strips of 8 MCU rows to share out
    assert sharded.rows_per_restart_unit(520, 1040) == 2
    assert sharded.rows_per_restart_unit(26, 40) == 20      # 40 MCUs and 26 per row meet again after 20 rows


# ---- progressive output, sharded (round 5): the orchestration of sharded.encode_step_progressive on CPU -------------------------
def _cut_progressive(jpg, geo, ri, W, H):
    """The oracle's whole-image progressive file taken apart: header_bytes[10] (what stands in front of each scan's data), and per scan
    the list of its restart intervals as byte strings (each with the RSTn that follows it; the last one has none)."""
    pos, hdrs, scans, start = 2, [], [], 0
    n = len(jpg)
    while pos < n:
        assert jpg[pos] == 0xFF, pos
        m = jpg[pos + 1]
        if m == 0xD9:
            break
        ln = (jpg[pos + 2] << 8) | jpg[pos + 3]
        pos += 2 + ln
        if m == 0xDA:
            hdrs.append(pos - start)
            a = pos
            ivs = []
            while True:           # entropy-coded data: up to the next marker that is neither a stuffed zero nor RSTn
                while not (jpg[pos] == 0xFF and jpg[pos + 1] != 0x00):
                    pos += 1
                if 0xD0 <= jpg[pos + 1] <= 0xD7:
                    pos += 2
                    ivs.append(bytes(jpg[a:pos]))
                    a = pos
                else:
                    ivs.append(bytes(jpg[a:pos]))
                    break
            scans.append(ivs)
            start = pos
    assert len(scans) == 10 and len(hdrs) == 10
    return hdrs, scans


class OracleProgressiveStrip:
    """Plays HipProgressiveStrip on CPU: rank `rank`'s restart intervals of every scan, cut out of the oracle's whole-image file (test
    infrastructure: what is under test is the offsets arithmetic, the one all-reduce and the transport)."""

    def __init__(self, O, W, H, q, css, ri, r0, r1, geo, rank):
        self.rank = rank
        jpg = O.encode_progressive(O.synth_rgb(W, H), q, css, ri)
        self.whole = jpg
        self.hdr_len, scans = _cut_progressive(jpg, geo, ri, W, H)
        self.hdr_bytes, pos = [], 0
        for i in range(10):
            self.hdr_bytes.append(jpg[pos:pos + self.hdr_len[i]])
            pos += self.hdr_len[i] + sum(len(b) for b in scans[i])
        hs, vs, mcux, mcuy = geo["hs"], geo["vs"], geo["mcux"], geo["mcuy"]
        script = [(3, 0), (1, 0), (1, 2), (1, 1), (1, 0), (1, 0), (3, 0), (1, 2), (1, 1), (1, 0)]      # (components, first component) of libjpeg's ten scans
        self.segs = []
        for i, (nc, c) in enumerate(script):
            if nc == 3:
                per_row, rows_all, k = mcux // ri, mcuy, 1
            elif c == 0:
                per_row, rows_all, k = (mcux * hs) // ri, (H + 7) // 8, vs
            else:
                per_row, rows_all, k = mcux // ri, ((H + vs - 1) // vs + 7) // 8, 1
            a, b = min(r0 * k, rows_all) * per_row, min(r1 * k, rows_all) * per_row
            assert (mcux * (hs if (nc == 1 and c == 0) else 1)) % ri == 0
            self.segs.append(b"".join(scans[i][a:b]))
        self.last = r1 == mcuy

    def statistics(self, stream=0):
        h = torch.zeros(10 * 4 * 257, dtype=torch.int32)
        h[0] = self.rank + 1
        self.hist = h
        return h

    def emit(self, stream=0):
        return [len(s) for s in self.segs], list(self.hdr_len)

    def place_into(self, buf, offsets, file_bytes, flags, stream=0):
        for i in range(10):
            buf[offsets[i]:offsets[i] + len(self.segs[i])] = torch.frombuffer(bytearray(self.segs[i]), dtype=torch.uint8)
            if flags & 1:
                buf[offsets[i] - self.hdr_len[i]:offsets[i]] = torch.frombuffer(bytearray(self.hdr_bytes[i]), dtype=torch.uint8)
        if flags & 2:
            buf[file_bytes - 2] = 0xFF
            buf[file_bytes - 1] = 0xD9

    def place_staged(self, sizes, stream=0):
        return [torch.frombuffer(bytearray(s), dtype=torch.uint8) if len(s) else torch.zeros(0, dtype=torch.uint8) for s in self.segs]


def _progressive_worker(rank, world, port, W, H, css, ri, q, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from nvjpeg_imagecompressor_amd import sharded
    geo = O.geometry(W, H, css)
    assert sharded.progressive_strip_interval(geo["mcux"]) == ri
    r0, r1 = sharded.partition_mcu_rows(geo["mcuy"], world, rank, 1)
    strip = OracleProgressiveStrip(O, W, H, q, css, ri, r0, r1, geo, rank) if r1 > r0 else None
    out = sharded.encode_step_progressive(torch, dist, strip, {})
    if strip is not None:       # the one collective summed every owner's statistics in place
        owners = sum(1 for r in range(world) if sharded.partition_mcu_rows(geo["mcuy"], world, r, 1)[1] > sharded.partition_mcu_rows(geo["mcuy"], world, r, 1)[0])
        assert int(strip.hist[0]) == owners * (owners + 1) // 2
    if rank == 0:
        open(out_path, "wb").write(out.numpy().tobytes())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("css,H", [(1, 250), (2, 250), (0, 64), (2, 64)])      # (2, 64): four MCU rows -- at world 8 four ranks own nothing
def test_progressive_n_rank_file_equals_one_rank_file(oracle, tmp_path, world, css, H):
    """Progressive output over N ranks (ImageCompressorImpl.cu:28 is the reference's own encoding): statistics of the ten scans in ONE
    all-reduce, sizes matrix all-gathered, segments placed scan by scan, ranks in order -- the file of the one-shot encoder, byte for
    byte. H = 250 is not a multiple of the MCU height (the last strip owns the clipped block rows of the single-component scans);
    (0, 64): 8 MCU rows for 8 ranks, one each."""
    W, q = 208, 92
    geo = oracle.geometry(W, H, css)
    ri = geo["mcux"]          # = progressive_strip_interval: 26 (4:4:4) or 13 MCUs per row, the interval must divide it
    out = tmp_path / "prog.jpg"
    if world == 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    mp.spawn(_progressive_worker, args=(world, _free_port(), W, H, css, ri, q, str(out)), nprocs=world, join=True)
    want = oracle.encode_progressive(oracle.synth_rgb(W, H), q, css, ri)
    assert out.read_bytes() == want


def test_progressive_offsets_arithmetic():
    from nvjpeg_imagecompressor_amd import sharded
    sizes = [[10 + r + i for i in range(10)] for r in range(3)]
    hdr = [100] + [30] * 9
    offs, total = sharded.progressive_offsets(sizes, hdr)
    assert offs[0][0] == 100 and offs[1][0] == 110 and offs[2][0] == 121
    assert offs[0][1] == 100 + (10 + 11 + 12) + 30
    assert total == sum(hdr) + sum(sum(r) for r in sizes) + 2
    assert sharded.progressive_strip_interval(520) == 520 and sharded.progressive_strip_interval(1040) == 520 and sharded.progressive_strip_interval(2080) == 520
    assert sharded.progressive_strip_interval(13) == 13
