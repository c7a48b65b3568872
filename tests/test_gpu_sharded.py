"""Strip sharding with the HIP strip encoder: N ranks share the one GPU of the test box (gloo carries the collectives;
RCCL refuses two ranks on one device), each encodes its strip. Two gathers: the device-side pipeline (sizes all-gathered
device to device, strips PUT into rank 0's buffer through hipIpc mappings -- between processes on one GPU here, between
GPUs over xGMI on the node) and the host-synchronised send/recv form. Must equal the oracle's one-shot file.
(At most 6 processes may use the test box's GPU at once: 5 ranks next to the test process is the largest world here; the
8-rank orchestration is covered on CPU, tests/test_sharded_cpu.py.)"""
import os
import socket
import sys
import zlib

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, W, H, css, optimize, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded
    torch.cuda.set_device(0)
    enc = sharded.make_hip_strip_encoder(torch, W, H, 95, optimize, css, rank, world, 0, "rgb")
    g = enc.geometry
    d_img = torch.empty((g["strip_rows"], W, 3), dtype=torch.uint8, device="cuda:0")
    mij.synth_image_device(d_img.data_ptr(), W, g["strip_y0"], g["strip_rows"], W * 3, bgr=False)
    torch.cuda.synchronize()
    strip = sharded.HipStripEncoder(torch, enc, d_img, "rgb")
    out = None
    for _ in range(2):   # twice: buffers are reused across steps
        out = sharded.encode_step(torch, dist, strip, optimize, {})
    if rank == 0:
        open(out_path, "wb").write(out.cpu().numpy().tobytes())
        open(out_path + ".ri", "w").write(str(g["restart_interval"]))
    dist.barrier()
    enc.close()
    dist.destroy_process_group()


def _noise_image(W, rows):
    """Noise in the right half (its blocks overflow the fast entropy coder's 16-word strips: the roomy coder and the handle's
    switch to 24-word strips get exercised inside the pipeline), the synthetic picture in the left."""
    import numpy as np
    from oracle import oracle as O
    full = O.synth_rgb(W, rows)
    full[:, W // 2:] = np.random.default_rng(77).integers(0, 256, (rows, W - W // 2, 3), dtype=np.uint8)
    return full


def _device_worker(rank, world, port, W, H, css, optimize, ri, nimg, collect_each, out_path, noise=False, backend="gloo", comms="ordered"):
    """sharded.DevicePipeline on real HIP handles: DEPTH images in flight, peer-mapped rank-0 buffers, k_put."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":       # RCCL, initialised the way bench.py does it (one rank: RCCL refuses two ranks on one device)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    whole, r0, r1 = sharded.strip_rows(W, H, 95, optimize, css, rank, world, ri)
    encs = [sharded.make_hip_strip_encoder(torch, W, H, 95, optimize, css, rank, world, 0, "rgb", restart_interval=ri) for _ in range(sharded.DEPTH)]
    strips = None
    imgs = []
    if encs[0] is not None:
        g = encs[0].geometry
        for i in range(nimg):       # image i = the synthetic image shifted down by 8 i rows (so that the strips differ per image)
            if noise:       # (the oracle is only the source of the test picture here, as in the CPU tests)
                y = g["strip_y0"] + 8 * i
                t = torch.from_numpy(_noise_image(W, H + 8 * nimg)[y:y + g["strip_rows"]].copy()).to(dev)
            else:
                t = torch.empty((g["strip_rows"], W, 3), dtype=torch.uint8, device=dev)
                mij.synth_image_device(t.data_ptr(), W, g["strip_y0"] + 8 * i, g["strip_rows"], W * 3, bgr=False)
            imgs.append(t)
        torch.cuda.synchronize()
        strips = [sharded.HipStripEncoder(torch, e, imgs[0], "rgb") for e in encs]
    else:
        assert r0 == r1 and rank > 0
    targets = sharded.open_file_targets(torch, dist, strips, rank, world, 0, whole)
    assert targets is not None, "hipIpc mapping of rank 0's buffers failed"
    pipe = sharded.DevicePipeline(torch, dist, strips, targets, optimize, device=dev, comms=comms)
    outs = []
    for i in range(nimg):
        if strips is not None:
            strips[i % sharded.DEPTH].d_img = imgs[i]
        pipe.step()
        if collect_each:
            o = pipe.collect()
            outs.append(None if o is None else o.cpu().numpy().tobytes())
    last = pipe.flush()
    if not collect_each:
        outs = [None] * (nimg - 1) + [None if last is None else last.cpu().numpy().tobytes()]
    for i, o in enumerate(outs):          # a file comes out on its image's root (the roots rotate over the ranks with a strip)
        if o is not None:
            open(out_path + ".%d" % i, "wb").write(o)
    if rank == 0:
        open(out_path + ".ri", "w").write(str(whole["restart_interval"]))
    dist.barrier()
    for e in encs:
        if e is not None:
            e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,css,optimize,ri,H,collect_each", [
    (2, 1, True, -1, 1000, True), (3, 2, True, -1, 1000, True), (5, 1, True, -1, 2504, True), (3, 0, False, -1, 1000, True),
    (3, 1, True, 260, 1000, True),     # interval = two MCU rows (130 MCUs per row): strips cut in units of two rows
    (5, 1, True, 3900, 480, True),     # one restart-aligned strip per 30 MCU rows: 2 units for 5 ranks, three of them own nothing
    (3, 1, True, -1, 1000, False), (5, 2, True, -1, 2504, False),
    (3, 1, True, -1, 1000, "noise"),   # half of every image is noise: intervals overflow the narrow fast coder (collected each step)
])
def test_device_pipeline_put_gather(oracle, tmp_path, world, css, optimize, ri, H, collect_each):
    import numpy as np
    W, nimg = 2080, 5
    out = str(tmp_path / "dev.jpg")
    noise = collect_each == "noise"
    collect_each = bool(collect_each)
    mp.spawn(_device_worker, args=(world, _free_port(), W, H, css, optimize, ri, nimg, collect_each, out, noise), nprocs=world, join=True)
    dri = int(open(out + ".ri").read())
    full = _noise_image(W, H + 8 * nimg) if noise else oracle.synth_rgb(W, H + 8 * nimg)
    for i in (range(nimg) if collect_each else [nimg - 1]):
        want = oracle.encode(np.ascontiguousarray(full[8 * i:8 * i + H]), 95, css, optimize, dri)
        got = open(out + ".%d" % i, "rb").read()
        assert len(got) == len(want) and zlib.crc32(got) == zlib.crc32(want), i


@pytest.mark.parametrize("comms", ["ordered", "per-slot"])
def test_device_pipeline_over_rccl_single_rank(oracle, tmp_path, comms):
    """The one thing a one-GPU box can show about the measured configuration: the pipeline's calls as RCCL takes them -- the
    eager communicator (every collective of every slot on it, in issue order: the default) or four split communicators, the
    int32 statistics all-reduce on a handle's own memory, the int64 sizes all-gather, the status all-reduce of `collect`, stream
    order between the library's kernels and RCCL's -- with a single rank (its own root), six images, four in flight, files
    against the oracle."""
    import numpy as np
    W, H, nimg = 2080, 1000, 6
    out = str(tmp_path / "rccl.jpg")
    mp.spawn(_device_worker, args=(1, _free_port(), W, H, 1, True, -1, nimg, False, out, False, "nccl", comms), nprocs=1, join=True)
    dri = int(open(out + ".ri").read())
    full = oracle.synth_rgb(W, H + 8 * nimg)
    want = oracle.encode(np.ascontiguousarray(full[8 * (nimg - 1):8 * (nimg - 1) + H]), 95, 1, True, dri)
    assert open(out + ".%d" % (nimg - 1), "rb").read() == want
    out2 = str(tmp_path / "rccl_each.jpg")
    mp.spawn(_device_worker, args=(1, _free_port(), W, H, 2, True, -1, 3, True, out2, False, "nccl", comms), nprocs=1, join=True)
    for i in range(3):
        want = oracle.encode(np.ascontiguousarray(full[8 * i:8 * i + H]), 95, 2, True, int(open(out2 + ".ri").read()))
        assert open(out2 + ".%d" % i, "rb").read() == want, i


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("css,optimize", [(1, True), (2, True), (0, False)])
def test_hip_strips_equal_one_shot(oracle, tmp_path, world, css, optimize):
    W, H = 2080, 1000    # 1000 rows: not a multiple of 16, the last strip owns the bottom edge
    out = tmp_path / "sharded.jpg"
    mp.spawn(_worker, args=(world, _free_port(), W, H, css, optimize, str(out)), nprocs=world, join=True)
    ri = int(open(str(out) + ".ri").read())
    want = oracle.encode(oracle.synth_rgb(W, H), 95, css, optimize, ri)
    got = out.read_bytes()
    assert len(got) == len(want) and zlib.crc32(got) == zlib.crc32(want)


def test_hip_strips_five_ranks_uneven(oracle, tmp_path):
    """Five ranks (the most the test box lets share its GPU next to the test process), strips of unequal height, bottom
    edge inside the last strip, 4:2:0 so that an MCU row is 16 pixels."""
    W, H, css, world = 4160, 2504, 2, 5
    out = tmp_path / "sharded5.jpg"
    mp.spawn(_worker, args=(world, _free_port(), W, H, css, True, str(out)), nprocs=world, join=True)
    ri = int(open(str(out) + ".ri").read())
    want = oracle.encode(oracle.synth_rgb(W, H), 95, css, True, ri)
    got = out.read_bytes()
    assert len(got) == len(want) and zlib.crc32(got) == zlib.crc32(want)


def test_reserved_output_buffer_is_uncached_and_still_exact(oracle):
    """mij_encoder_reserve_output (what every strip-owning rank calls before it exports its buffer) allocates device-uncached
    memory: peers write into that buffer behind this GPU's caches, so none of its lines may live in them.
    Header (K3), compaction (K6), the device-side result and the copy to the host all work on it unchanged; MIJ_SHARED_OUT=cached
    keeps plain memory (child process: the switch is read per call, the assertion is the point)."""
    import subprocess
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded
    W, H = 1040, 512
    img = oracle.synth_rgb(W, H)
    with mij.Encoder(W, H, 92, True, 1) as enc:
        assert not enc.output_is_uncached()
        enc.reserve_output(8 * sharded.full_scan_capacity(enc.geometry))
        assert enc.output_is_uncached()
        h = mij.encoder.ipc_export(enc.output_buffer()[0])        # exportable: what open_file_targets does next
        assert len(h) == 64
        for _ in range(3):
            got = enc.encode_host(img, "rgb")
            assert got == oracle.encode(img, 92, 1, True, enc.geometry["restart_interval"])
    code = ("import sys; sys.path.insert(0, %r); import nvjpeg_imagecompressor_amd as mij\n"
            "e = mij.Encoder(1040, 512, 92, True, 1); e.reserve_output(1 << 24); assert not e.output_is_uncached(); print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MIJ_SHARED_OUT="cached"), timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def _bench(args, env_extra=None, launcher=False, timeout=900):
    import json
    import subprocess
    env = dict(os.environ, MIJ_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MIJ_BENCH_CHILD", "MIJ_BENCH_DIRECT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher:
        port = str(_free_port())
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher), "--master-addr", "127.0.0.1", "--master-port", port]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr


@pytest.fixture(scope="module")
def one_gpu_line():
    d, _ = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-psnr", "--height", "4000"])
    assert d["n_gpus"] == 1 and "launcher" not in d          # N = 1 runs in the process of the command line, as ever
    # round 4: the loop behind `value` overlaps two images on two streams; per-kernel figures come from a separate one-stream pass
    assert d["loop"] == "overlap" and d["config"]["streams"] == 2 and d["config"]["images_in_flight"] == 3
    assert d["per_kernel_pass"]["file_identical_to_timed_loop"] is True and d["per_kernel_pass"]["streams"] == 1
    assert d["roofline"]["stage_A_alone"]["avg_launch_ms"] > 0 and 0 < d["roofline"]["stage_A_alone"]["frac"] < 1
    assert d["roofline"]["stage_A_alone"]["avg_launch_ms"] < d["roofline"]["avg_launch_ms"]        # no statistics: the shorter kernel
    assert len(d["clock"]["probes"]) >= 5 and all(500 < p["valu_MHz"] < 3000 for p in d["clock"]["probes"])
    assert d["single_image_ms"] > 0
    return d


@pytest.mark.parametrize("gather,comms", [("put", "ordered"), ("put", "per-slot"), ("sendrecv", None), ("serial", None)])
def test_bench_multi_rank_rehearsal(one_gpu_line, gather, comms):
    """PLAIN `python bench.py --gpus 3` -- no launcher in the command -- end to end (self-launched fresh ranks, strips,
    collectives, gather, the JSON line), rehearsed with three ranks on the one GPU over gloo: the file must be the single-GPU
    file. All forms of the step: four images in flight with the put gather and a rotating root (sharded.DevicePipeline, on
    one ordered communicator -- the default -- and on one communicator per slot), two in flight with send/recv
    (sharded.StripPipeline) and one at a time (sharded.encode_step)."""
    d, _ = _bench(["--gpus", "3", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--height", "4000", "--gather", gather]
                  + (["--comms", comms] if comms else []))
    assert d["n_gpus"] == 3 and d["steps"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"] == "strips3" and d["config"]["images_in_flight"] == {"put": 4, "sendrecv": 2, "serial": 1}[gather]
    assert d["config"]["gather"] == gather and d["gather_fallback"] is None and d["launcher"].startswith("self: 3 fresh child ranks")
    assert d["config"]["comms"] == comms
    assert d["jpeg_crc32"] == one_gpu_line["jpeg_crc32"] and d["jpeg_bytes"] == one_gpu_line["jpeg_bytes"] and abs(d["psnr_db"] - 31.1) < 0.3
    assert d["library_source_hash"] == one_gpu_line["library_source_hash"]
    assert d["single_image_latency_ms"] > 0 and d["rccl_ranks"] is None and "gloo" in d["collective_backend"]
    if gather == "put":
        # one image per root before the timed region and one per root after it, all equal to the timed region's file
        assert d["files_verified"]["roots"] == 3 and d["files_verified"]["images"] == 6 and d["files_verified"]["identical_to_timed_file"] is True
        assert d["put_GB/s"]["samples"] >= 2 and d["put_GB/s"]["min"] > 0
        # round 5: what the root takes in per image, so that a first multi-GPU run explains itself
        ri = d["root_inbound"]
        assert len(ri["strip_bytes_per_rank"]) == 3 and sum(ri["strip_bytes_per_rank"]) > 0
        assert ri["inbound_bytes_per_image"] == sum(ri["strip_bytes_per_rank"]) - ri["strip_bytes_per_rank"][ri["root"]]
    assert sorted(d["per_rank_total_ms"]) == ["0", "1", "2"]
    if d["slowest_rank"] is not None:
        assert d["slowest_rank"]["stage_ms"]["total"] == max(v for v in d["per_rank_total_ms"].values() if v)


def test_bench_under_an_external_launcher(one_gpu_line):
    """The driver's documented command shape: torch.distributed.run starts N ranks of bench.py. Its local rank 0 supervises N
    fresh children (the same ones the plain command starts), the launcher's other ranks leave at once."""
    d, _ = _bench(["--gpus", "3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-psnr", "--height", "4000"], launcher=3)
    assert d["n_gpus"] == 3 and d["launcher"].startswith("external launcher (3 ranks)")
    # the default gather is `auto`: a complete run with the put pipeline AND one with send/recv, the faster one's line printed
    assert set(d["gather_runs"]) == {"put", "sendrecv"} and all(r["value"] > 0 for r in d["gather_runs"].values())
    assert d["config"]["gather"] == max(d["gather_runs"], key=lambda k: d["gather_runs"][k]["value"])
    assert d["jpeg_crc32"] == one_gpu_line["jpeg_crc32"]


@pytest.mark.parametrize("inject,expect", [("die:put:1:pipeline", "put: rank 1 exited with status 3"),
                                           ("hang:put:2:pipeline", "put: no progress for")])
def test_bench_falls_back_to_send_recv_with_fresh_ranks(one_gpu_line, inject, expect):
    """A rank of the put pipeline dies / stalls: the supervisor ends the ranks it started and starts fresh ones with send/recv;
    the line says why, and the file is still the single-GPU file."""
    d, err = _bench(["--gpus", "3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-psnr", "--height", "4000"],
                    env_extra={"MIJ_BENCH_INJECT": inject, "MIJ_BENCH_WATCHDOG_S": "45"})
    assert d["config"]["gather"] == "sendrecv" and expect in d["gather_fallback"], d["gather_fallback"]
    assert d["jpeg_crc32"] == one_gpu_line["jpeg_crc32"] and "starting fresh ranks" in err


# ---- progressive output over N ranks (round 5) --------------------------------------------------------------------------------------
def _progressive_worker(rank, world, port, W, H, css, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    enc = sharded.make_hip_strip_encoder(torch, W, H, 95, True, css, rank, world, 0, "rgb", progressive=True)
    strip = None
    if enc is not None:
        g = enc.geometry
        d_img = torch.empty((g["strip_rows"], W, 3), dtype=torch.uint8, device=dev)
        mij.synth_image_device(d_img.data_ptr(), W, g["strip_y0"], g["strip_rows"], W * 3, bgr=False)
        torch.cuda.synchronize()
        strip = sharded.HipProgressiveStrip(torch, enc, d_img, "rgb")
    out = None
    cache = {"device": dev}
    for _ in range(2):          # twice: buffers, statistics and tables are reused across images
        out = sharded.encode_step_progressive(torch, dist, strip, cache, torch.cuda.current_stream().cuda_stream)
    if rank == 0:
        open(out_path, "wb").write(out.cpu().numpy().tobytes())
        open(out_path + ".ri", "w").write(str(enc.geometry["restart_interval"]))
    dist.barrier()
    if enc is not None:
        enc.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,css,H", [(2, 1, 1000), (3, 2, 1000), (5, 1, 2504), (3, 0, 488), (4, 1, 250)])
def test_progressive_n_ranks_equal_the_one_shot_file(oracle, tmp_path, world, css, H):
    """The reference's own encoding (progressive, ImageCompressorImpl.cu:28) over N ranks with HIP strips: ten scans' statistics in one
    all-reduce, ten segments per rank placed scan by scan -- byte-identical to the oracle's (= libjpeg's) one-shot file with the same
    restart interval, and to what ONE rank writes through mij_encode_entropy. (4, 1, 250): the last strip owns clipped block rows and a
    bottom edge; (3, 0, 488): 4:4:4.)"""
    import numpy as np
    W = 2080
    out = tmp_path / "prog.jpg"
    mp.spawn(_progressive_worker, args=(world, _free_port(), W, H, css, str(out)), nprocs=world, join=True)
    ri = int(open(str(out) + ".ri").read())
    want = oracle.encode_progressive(oracle.synth_rgb(W, H), 95, css, ri)
    got = out.read_bytes()
    assert len(got) == len(want) and got == want
    import nvjpeg_imagecompressor_amd as mij
    d = torch.from_numpy(oracle.synth_rgb(W, H)).cuda()
    with mij.Encoder(W, H, 95, True, css, restart_interval=ri, progressive=True) as enc:       # one rank, the whole-image path
        enc.encode_device(d.data_ptr(), W * 3, "rgb")
        assert enc.retrieve() == want
    with mij.Encoder(W, H, 95, True, css, restart_interval=ri, progressive=True) as enc:       # one rank through the strip protocol
        enc.transform(d.data_ptr(), W * 3, "rgb")
        enc.prog_statistics()
        sizes, hdr = enc.prog_emit()
        from nvjpeg_imagecompressor_amd import sharded
        offs, total = sharded.progressive_offsets([sizes], hdr)
        enc.prog_place(offs[0], 0, 0, total, 3)
        assert enc.retrieve() == want


def test_progressive_strip_needs_an_interval_that_divides_the_row(mij):
    with pytest.raises(mij.MiJpegError):
        mij.Encoder(2080, 1000, 95, True, 1, restart_interval=64, strip_mcu_row0=0, strip_mcu_rows=8, progressive=True)     # 130 MCUs per row
    with mij.Encoder(2080, 1000, 95, True, 1, restart_interval=65, strip_mcu_row0=8, strip_mcu_rows=8, progressive=True) as e:
        assert e.geometry["strip_y0"] == 64


def test_progressive_fullsize_three_ranks_by_crc(tmp_path):
    """BASELINE's image in the reference's own encoding over three ranks (one GPU here, gloo): the committed fingerprint of the CPU
    oracle's one-shot file at the sharded interval (tests/golden/big_8320x40000_q95.json, css1_ri520_progressive)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "big_8320x40000_q95.json")))["cases"]["css1_ri520_progressive"]
    out = tmp_path / "prog_full.jpg"
    mp.spawn(_progressive_worker, args=(3, _free_port(), 8320, 40000, 1, str(out)), nprocs=3, join=True)
    assert int(open(str(out) + ".ri").read()) == 520
    got = out.read_bytes()
    assert (len(got), "%08x" % zlib.crc32(got)) == (gold["len"], gold["crc32"])
