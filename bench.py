#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mpixels/s JPEG encode of 8320x40000 RGB8, q95, 4:2:2, optimised
Huffman (reference README.md:48 config), device-resident input -> device-resident JFIF bitstream.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run, one rank/GPU)

A "step" is one whole encode of the image. At N > 1 the image is cut into restart-interval-aligned strips of MCU
rows, one per rank: transform+statistics locally, ONE all-reduce of the 4x257 symbol statistics
(RCCL), entropy coding locally, all-gather of strip sizes, gather of strip bitstreams to rank 0. Total work is fixed
as N grows => "scaling": "strong".  Rank 0 prints ONE JSON line.
"""
import argparse
import io
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG, QUALITY, CSS_NAME = 8320, 40000, 95, "422"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=W_IMG)
    ap.add_argument("--height", type=int, default=H_IMG)
    ap.add_argument("--css", default=CSS_NAME)
    ap.add_argument("--quality", type=int, default=QUALITY)
    ap.add_argument("--no-optimize", action="store_true")
    ap.add_argument("--fmt", default="bgr", choices=["bgr", "rgb"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1000, help="rows per CPU-baseline strip (one strip per core)")
    return ap.parse_args()


def algorithmic_bytes_per_pixel(css, ratio):
    """stage A reads 3 B/px and writes int16 coefficients 2(1+f); B reads them; C reads them and writes
    the bitstream (3*ratio B/px)."""
    f = {"444": 2.0, "422": 1.0, "440": 1.0, "420": 0.5, "411": 0.5, "410": 0.25}[css]
    return {"transform": 3 + 2 * (1 + f), "statistics": 2 * (1 + f), "entropy": 2 * (1 + f) + 3 * ratio,
            "compact": 2 * 3 * ratio}


def _cpu_share():
    """CPUs this job may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baselines(args, optimize):
    """Bounded CPU sample of the same workload on the host cores: (a) the oracle C port, (b) libjpeg-turbo via Pillow.
    One strip of the synthetic image per core, encoded concurrently (both release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from PIL import Image
    # host cores this process may use (the GPU box gives a 1-GPU job a share of the node, not all of os.cpu_count())
    cores = int(os.environ.get("MIJ_BENCH_CORES", "0")) or _cpu_share()
    rows = args.cpu_sample_rows
    css = {"444": 0, "422": 1, "420": 2, "440": 3, "411": 4, "410": 5}[args.css]
    strips = [O.synth_rgb(args.width, args.height, y0=(i * rows) % max(1, args.height - rows), rows=rows) for i in range(cores)]
    mpix = cores * rows * args.width / 1e6

    def run_oracle(s):
        return len(O.encode(s, args.quality, css, optimize, 104))

    def run_turbo(s):
        b = io.BytesIO()
        kw = {}
        if css <= 2:
            Image.fromarray(s).save(b, "JPEG", quality=args.quality, subsampling=css, optimize=optimize, **kw)
            return len(b.getvalue())
        return 0

    out = {}
    for name, fn in (("port", run_oracle), ("turbo", run_turbo)):
        if name == "turbo" and css > 2:
            continue
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                sizes = list(ex.map(fn, strips))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name] = {"value": round(mpix / best, 2), "unit": "Mpixels/s", "cores": cores,
                     "sample": "%d strips of %dx%d synthetic RGB8 (one per core, concurrent), q%d %s %s, best of 2" %
                               (cores, args.width, rows, args.quality, args.css, "optimised" if optimize else "fixed"),
                     "bytes": int(sum(sizes))}
    return out


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import nvjpeg_imagecompressor_amd as mij

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    optimize = not args.no_optimize
    W, H = args.width, args.height

    # ---- strip partition (pure arithmetic, no communication) -------------------------------------------------
    probe = mij.Encoder(W, H, args.quality, optimize, args.css, device=local_rank, strip_mcu_row0=0, strip_mcu_rows=1)
    g0 = probe.geometry
    probe.close()
    mcu_rows, mcu_h = g0["mcu_rows"], g0["mcu_h"]
    r0 = rank * mcu_rows // world
    r1 = (rank + 1) * mcu_rows // world
    enc = mij.Encoder(W, H, args.quality, optimize, args.css, device=local_rank, strip_mcu_row0=r0, strip_mcu_rows=r1 - r0)
    geo = enc.geometry
    y0, rows = geo["strip_y0"], geo["strip_rows"]
    pitch = W * 3
    d_img = torch.empty((rows, W, 3), dtype=torch.uint8, device=dev)
    mij.synth_image_device(d_img.data_ptr(), W, y0, rows, pitch, bgr=(args.fmt == "bgr"),
                           stream=torch.cuda.current_stream().cuda_stream)
    d_hist = torch.zeros(4 * 257, dtype=torch.int32, device=dev)
    enc.set_histogram_buffer(d_hist.data_ptr())
    enc.enable_timing(True)
    torch.cuda.synchronize()

    gathered = {}
    stage_acc = {}

    def step(record):
        s = torch.cuda.current_stream().cuda_stream
        enc.transform(d_img.data_ptr(), pitch, args.fmt, 0, s)
        if world > 1 and optimize:
            dist.all_reduce(d_hist)                      # the only data-path collective before entropy coding
        enc.entropy(s)
        res = enc.result()                               # waits for this rank's strip; sizes now known on the host
        if record:
            for k, v in enc.stage_times().items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
        if world > 1:
            sizes = torch.zeros(world, dtype=torch.int64, device=dev)
            mine = torch.tensor([res["scan_bytes"]], dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sizes, mine)
            sz = sizes.cpu().tolist()
            view = _device_bytes(torch, res["d_buffer"] + res["scan_offset"], res["scan_bytes"], dev)
            if rank == 0:
                total = res["header_bytes"] + sum(sz)
                buf = gathered.get("buf")
                if buf is None or buf.numel() < total:
                    buf = torch.empty(total + (total >> 3), dtype=torch.uint8, device=dev)
                    gathered["buf"] = buf
                hdr = _device_bytes(torch, res["d_buffer"] + res["header_offset"], res["header_bytes"] + sz[0], dev)
                buf[:hdr.numel()].copy_(hdr)
                ops, off = [], hdr.numel()
                for r in range(1, world):
                    ops.append(dist.P2POp(dist.irecv, buf[off:off + sz[r]], r))
                    off += sz[r]
                if ops:
                    for w_ in dist.batch_isend_irecv(ops):
                        w_.wait()
                gathered["len"] = total
            else:
                for w_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, view, 0)]):
                    w_.wait()
        else:
            gathered["res"] = res
        return res

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(True)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    mpix = W * H / 1e6
    value = mpix / (ms_per_step / 1e3)

    # ---- rank 0: assemble, verify, report ------------------------------------------------------------------------
    out = None
    if rank == 0:
        if world > 1:
            jpeg = gathered["buf"][:gathered["len"]].cpu().numpy().tobytes()
        else:
            jpeg = enc.retrieve()
        ratio = len(jpeg) / (3.0 * W * H)
        stages = {k: v / args.steps for k, v in stage_acc.items()}
        strip_px = rows * W
        bpp = algorithmic_bytes_per_pixel(args.css, ratio)
        stage_roof = {}
        for k in ("transform", "statistics", "entropy", "compact"):
            if stages.get(k, 0) > 0:
                gbs = bpp[k] * strip_px / (stages[k] * 1e-3) / 1e9
                stage_roof[k] = {"ms": round(stages[k], 4), "GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        dom = max(stage_roof, key=lambda k: stage_roof[k]["ms"]) if stage_roof else None
        kname = {"transform": "k_transform", "statistics": "k_histogram", "entropy": "k_encode", "compact": "k_compact"}
        roofline = None
        if dom:
            roofline = {"bound": "hbm", "kernel": kname[dom], "achieved": stage_roof[dom]["GB/s"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": stage_roof[dom]["frac"], "traffic": None,
                        "algorithmic_bytes_per_launch": int(bpp[dom] * strip_px), "avg_launch_ms": stage_roof[dom]["ms"]}
        out = {
            "metric": "Mpixels/s encode (+PSNR, ratio) 8320x40000 q95 4:2:2 @1/2/4/8 GPU",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d RGB8 (%s interleaved, device resident) -> baseline JFIF, q%d, 4:%s:%s, %s Huffman, "
                                   "DRI=%d MCUs" % (W, H, args.fmt.upper(), args.quality, args.css[1], args.css[2],
                                                    "optimised" if optimize else "fixed", geo["restart_interval"]),
                       "parallelism": "strips%d" % world, "restart_interval": geo["restart_interval"]},
            "jpeg_bytes": len(jpeg), "ratio": round(ratio, 5), "jpeg_crc32": "%08x" % zlib.crc32(jpeg),
            "roofline": roofline, "stage_roofline": stage_roof, "stage_ms": {k: round(v, 4) for k, v in stages.items()},
        }
        if not args.no_psnr:
            out["psnr_db"], out["psnr_note"] = _psnr_check(jpeg, W, H, args.fmt, d_img if world == 1 else None)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baselines(args, optimize)
            out["cpu_baseline"] = dict(cb["port"], kind="port", impl="oracle/jpeg_oracle.c")
            if "turbo" in cb:
                out["cpu_libjpeg_turbo"] = dict(cb["turbo"], impl="libjpeg-turbo via Pillow")
        print(json.dumps(out), flush=True)
    enc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _device_bytes(torch, ptr, nbytes, dev):
    """uint8 tensor view over device memory owned by libmijpeg (valid until the next encode on the handle)."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=dev)


def _psnr_check(jpeg, W, H, fmt, d_img):
    """Decode with a stock decoder (libjpeg-turbo via Pillow) and compare with the source image, in bands."""
    import numpy as np
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    t0 = time.perf_counter()
    dec = np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))
    from oracle import oracle as O
    se, band = 0.0, 2000
    for y in range(0, H, band):
        n = min(band, H - y)
        if d_img is not None:
            src = d_img[y:y + n].cpu().numpy()
            if fmt == "bgr":
                src = src[..., ::-1]
        else:
            src = O.synth_rgb(W, H, y, n)
        diff = dec[y:y + n].astype(np.int32) - src.astype(np.int32)
        se += float((diff * diff).sum())
    mse = se / (3.0 * W * H)
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    return round(psnr, 3), "stock decoder: Pillow/libjpeg-turbo, full image, %.1fs" % (time.perf_counter() - t0)


if __name__ == "__main__":
    main()
