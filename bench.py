#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mpixels/s JPEG encode of 8320x40000 RGB8, q95, 4:2:2, optimised
Huffman (reference README.md:48 config), device-resident input -> device-resident JFIF bitstream.

  python bench.py [--gpus N --steps K --warmup W]

N > 1 launches ITSELF: the process started by the command line never touches a GPU; it starts N fresh child processes (one
rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, RCCL between them), relays rank 0's JSON line
and exits with their status. It watches them: if a rank dies or the run makes no progress within a bound, the children are
killed (exactly the process groups it started) and FRESH ones are started with the next gather of the ladder
put -> sendrecv -> serial; the line then carries "gather_fallback": "<reason>". Started under an external launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) local rank 0 becomes that supervisor and the
launcher's other ranks leave at once, so both command shapes run the same children with the same safety net
(MIJ_BENCH_DIRECT=1: be a plain rank of the external launcher instead, no supervisor).

A "step" is one whole encode of the image. On one GPU (round 4) the timed loop runs the transform of image i on one HIP stream
and the entropy stage (tables, Huffman coding, size scan, stuffing + compaction) of image i-1 on a second one, three handles taking
turns, so the issue-bound transform and the latency-bound coder share the machine (--loop overlap, the default; --loop
tables-ahead / one-stream / two-streams are the earlier rounds' loops). Everything the line says about a SINGLE kernel -- `roofline`,
`stage_roofline`, `stage_ms` -- comes from a short separate pass right AFTER the timed region in which every kernel of an image runs alone
on one stream, labelled as such, followed by a stage-A-alone pass (the transform without the fused statistics, the kernel the
north-star's 70 % is about). The shader clock needs tens of milliseconds of load to settle, more than W = 5 warm-up steps last: behind
the warm-up steps the same loop runs on, untimed, in rounds of ten steps until two clock readings agree (clock.settle_steps), and
`clock` records what the clock was right before and right after the timed region and after each pass. At N > 1 the image is cut into restart-interval-aligned strips of MCU
rows, one per rank (nvjpeg_imagecompressor_amd/sharded.py): transform+statistics locally, ONE all-reduce
of the 4x257 symbol statistics (RCCL), entropy coding locally, all-gather of strip sizes (device to device), and every
rank PUTS its strip into the file the image's root assembles (peer-mapped buffer, one xGMI link per rank; the root rotates
over the ranks from image to image, so no GPU's inbound links carry every file) at the offset a
kernel derives from the gathered sizes; four images in flight per rank, no host wait in a step (sharded.DevicePipeline;
RCCL send/recv with host-side sizes if the peer mapping is unavailable or the put pipeline fails). Total work is fixed as N grows => "scaling":
"strong".  Rank 0 prints ONE JSON line.
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
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL and the peer-mapped gather buffers need on this host driver

W_IMG, H_IMG, QUALITY, CSS_NAME = 8320, 40000, 95, "422"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=W_IMG)
    ap.add_argument("--height", type=int, default=H_IMG)
    ap.add_argument("--css", default=CSS_NAME)
    ap.add_argument("--quality", type=int, default=QUALITY)
    ap.add_argument("--no-optimize", action="store_true")
    ap.add_argument("--progressive", action="store_true", help="SOF2 output (the reference's nvJPEG setting, ImageCompressorImpl.cu:28); not the headline config. "
                    "N > 1: strips with a restart interval that divides the MCU row, one image at a time (sharded.encode_step_progressive)")
    ap.add_argument("--fmt", default="bgr", choices=["bgr", "rgb"])
    ap.add_argument("--restart-interval", type=int, default=-1, help="DRI in MCUs; -1 = the library's automatic choice (the headline config)")
    ap.add_argument("--loop", default=None, choices=["overlap", "tables-ahead", "one-stream", "two-streams"],
                    help="one GPU, the loop behind `value`. overlap (default): transform of image i on one stream, entropy stage of image i-1 on "
                         "another, three handles; tables-ahead (round 3): all wide kernels on one stream, the one-workgroup table build on a "
                         "side stream; one-stream (rounds 1-2): two handles, every kernel in issue order on one stream; two-streams: two "
                         "handles, each image entirely on its own stream. The per-kernel figures of the line always come from the separate "
                         "one-stream pass, whatever the loop")
    ap.add_argument("--two-streams", action="store_true", help=argparse.SUPPRESS)           # = --loop two-streams (round 2-3 spelling)
    ap.add_argument("--no-tables-ahead", dest="tables_ahead", action="store_false", help=argparse.SUPPRESS)   # = --loop one-stream
    ap.add_argument("--tables-ahead", dest="tables_ahead", action="store_true", help=argparse.SUPPRESS)       # = --loop tables-ahead
    ap.set_defaults(tables_ahead=None)
    ap.add_argument("--settle-rounds", type=int, default=8, help="one GPU: at most this many untimed rounds of ten steps of the timed loop behind the "
                    "warm-up steps, ended as soon as two successive shader-clock readings agree within 1 %% (0 = none); reported as clock.settle_steps")
    ap.add_argument("--kernel-pass", type=int, default=10, help="one GPU: images of the separate one-stream pass the per-kernel times come from")
    ap.add_argument("--stage-a-pass", type=int, default=10, help="one GPU: launches of the transform WITHOUT statistics (stage A alone) timed "
                    "after the kernel pass (0 = skip)")
    ap.add_argument("--fixed-root", action="store_true", help="N > 1, put gather: rank 0 assembles every file (default: the assembling rank "
                    "rotates from image to image, so that the strips of consecutive images arrive over different GPUs' links)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="one GPU: skip the sections behind the headline passes (`configs.table1` = BASELINE config 3, "
                    "`configs.secondary` = config 5, `decode` = both decoders); they only run on the default workload anyway")
    ap.add_argument("--no-progressive-decode", action="store_true", help="skip `decode.progressive_nodri` (Pillow writes the 190-MB progressive file "
                    "on the host first: ~9 s)")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1000, help="rows per CPU-baseline strip (one strip per core)")
    ap.add_argument("--cpu-one-core-rows", type=int, default=0, help="rows of the 1-core CPU sample (0 = the whole image, SURVEY 8d (i))")
    ap.add_argument("--gather", default=os.environ.get("MIJ_SHARDED_GATHER", "auto"), choices=["auto", "put", "sendrecv", "serial"],
                    help="N > 1: put = strips written into the assembling rank's peer-mapped buffer, sizes stay on the device; "
                         "sendrecv = host-side sizes + RCCL send/recv, two images in flight; serial = one image at a time. The supervisor "
                         "falls back along put -> sendrecv -> serial with fresh processes when a run dies or stalls. auto (default) = that "
                         "ladder from put, and -- because no multi-GPU run has ever told which gather a node prefers -- a second complete run "
                         "with send/recv afterwards: the line printed is the faster one's, both values are in `gather_runs`")
    ap.add_argument("--comms", default="ordered", choices=["ordered", "per-slot"],
                    help="N > 1, put gather: ordered = every collective on ONE communicator in one global order, an image's all-gather issued "
                         "behind the next image's all-reduce (default); per-slot = one communicator per image slot (experiment: concurrent "
                         "communicators are not ordered against each other across ranks)")
    ap.add_argument("--no-fallback", action="store_true", help="N > 1: fail instead of retrying with the next gather of the ladder")
    return ap.parse_args()


def algorithmic_bytes_per_pixel(css, ratio):
    """stage A reads 3 B/px and writes int16 coefficients 2(1+f); the entropy coder reads them and
    writes the bitstream (3*ratio B/px); stuffing+compaction reads and writes the bitstream."""
    f = {"444": 2.0, "422": 1.0, "440": 1.0, "420": 0.5, "411": 0.5, "410": 0.25}[css]
    return {"transform": 3 + 2 * (1 + f), "entropy": 2 * (1 + f) + 3 * ratio, "compact": 2 * 3 * ratio}


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


def cpu_baselines(args, optimize, restart_interval):
    """Bounded CPU sample of the same workload on the host cores (tools/cpu_baseline.py): libjpeg-turbo via
    Pillow on all cores and on one core, the oracle C port, and IJG libjpeg 9d for 4:4:0 / 4:1:1. Runs in a separate
    interpreter so that its process pool never forks a process that holds a HIP context."""
    import subprocess
    cores = int(os.environ.get("MIJ_BENCH_CORES", "0")) or _cpu_share()
    css = {"444": 0, "422": 1, "420": 2, "440": 3, "411": 4, "410": 5}[args.css]
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--width", str(args.width), "--height", str(args.height),
           "--rows", str(args.cpu_sample_rows), "--one-core-rows", str(args.cpu_one_core_rows), "--cores", str(cores),
           "--quality", str(args.quality), "--css", str(css),
           "--optimize", str(int(optimize)), "--ri", str(restart_interval)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if r.returncode:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-500:])
    return json.loads(r.stdout.strip().splitlines()[-1])


# =====================================================================================================================
# Supervisor: the process of `python bench.py --gpus N` (N > 1). It never imports torch or touches a GPU.
# =====================================================================================================================
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _child_argv(mode):
    """This command line with the gather replaced by `mode`."""
    out, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a == "--gather":
            skip = True
            continue
        if a.startswith("--gather="):
            continue
        out.append(a)
    return [sys.executable, os.path.abspath(__file__)] + out + ["--gather", mode]


def _kill_children(procs):
    """Ends exactly the processes this supervisor started (each child leads its own process group): TERM, then KILL."""
    import signal
    for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 10.0)):
        alive = [p for p in procs if p.poll() is None]
        if not alive:
            return
        for p in alive:
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.monotonic() + wait
        while time.monotonic() < t_end and any(p.poll() is None for p in alive):
            time.sleep(0.1)


def _run_attempt(world, mode):
    """One set of fresh child ranks. Returns (json_line or None, failure reason or None, note)."""
    import queue
    import subprocess
    import threading
    init_bound = float(os.environ.get("MIJ_BENCH_WATCHDOG_INIT_S", "420"))     # first import of torch on a fresh box: minutes
    step_bound = float(os.environ.get("MIJ_BENCH_WATCHDOG_S", "180"))          # between two progress marks afterwards
    port = _free_port()
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k not in ("GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                                                                        "GROUP_WORLD_SIZE", "ROLE_NAME")}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MIJ_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("MIJ_BENCH_NO_PIPELINE", None)
        procs.append(subprocess.Popen(_child_argv(mode), env=env, cwd=ROOT, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else 2, text=(r == 0) or None))       # other ranks: stdout -> our stderr (fd 2)
    _LIVE[:] = procs                       # what a SIGTERM / SIGINT to the supervisor must take down with it (the children lead their own sessions)
    lines = queue.Queue()

    def pump():
        for ln in procs[0].stdout:
            lines.put(ln.rstrip("\n"))
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    result, phase, last, reason = None, "start", time.monotonic(), None
    while True:
        try:
            while True:
                ln = lines.get(timeout=0.25)
                if ln.startswith("##progress"):
                    phase, last = ln[len("##progress"):].strip(), time.monotonic()
                elif ln.startswith("{"):
                    result, last = ln, time.monotonic()
                elif ln:
                    print(ln, file=sys.stderr, flush=True)
        except queue.Empty:
            pass
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and result is None:
            reason = "rank %d exited with status %d after '%s'" % (bad[0][0], bad[0][1], phase)
            break
        if all(c is not None for c in codes):
            if result is None:
                reason = "all ranks exited without a result line (after '%s')" % phase
            elif bad:          # measured, verified and reported by rank 0 with every rank taking part; a rank then failed on its way out
                _kill_children(procs)
                return result, None, "rank %d exited with status %d after the result line" % bad[0]
            break
        bound = init_bound if phase == "start" else step_bound
        if time.monotonic() - last > bound:
            if result is not None:       # measured and reported; a rank is stuck tearing down
                _kill_children(procs)
                return result, None, "ranks still alive %.0f s after the result line were ended by the supervisor" % bound
            reason = "no progress for %.0f s after '%s'" % (bound, phase)
            break
    _kill_children(procs)
    th.join(timeout=2.0)
    return (result, None, None) if reason is None else (None, reason, None)


_LIVE = []


def _on_signal(signum, frame):
    _kill_children(list(_LIVE))
    sys.exit(128 + signum)


def supervise(args, world, launcher):
    import signal
    signal.signal(signal.SIGTERM, _on_signal)
    signal.signal(signal.SIGINT, _on_signal)
    auto = args.gather == "auto"
    ladder = {"auto": ["put", "sendrecv", "serial"], "put": ["put", "sendrecv", "serial"], "sendrecv": ["sendrecv", "serial"], "serial": ["serial"]}[args.gather]
    if args.no_fallback or args.progressive:
        ladder = ladder[:1]
    failures = []
    for mode in ladder:
        line, reason, note = _run_attempt(world, mode)
        if line is not None:
            out = json.loads(line)
            out["launcher"] = launcher
            out["gather_fallback"] = "; ".join(failures) if failures else None
            if note:
                out["teardown_note"] = note
            if auto and mode == "put" and not args.no_fallback and os.environ.get("MIJ_BENCH_NO_AB") != "1":
                # The put pipeline ran. Which gather a real node prefers has never been measured, so the send/recv pipeline gets a
                # complete run of its own (fresh ranks); a failure there costs nothing. The faster run's line is the one printed.
                runs = {"put": {"value": out["value"], "ms_per_step": out.get("ms_per_step")}}
                line2, reason2, note2 = _run_attempt(world, "sendrecv")
                if line2 is not None:
                    out2 = json.loads(line2)
                    runs["sendrecv"] = {"value": out2["value"], "ms_per_step": out2.get("ms_per_step")}
                    if out2["value"] > out["value"] and out2.get("jpeg_crc32") == out.get("jpeg_crc32"):
                        out2["launcher"], out2["gather_fallback"] = launcher, None
                        if note2:
                            out2["teardown_note"] = note2
                        out = out2
                else:
                    runs["sendrecv"] = {"failed": reason2}
                out["gather_runs"] = runs
                out["gather_runs_note"] = "two complete runs with fresh ranks, one per gather; this line is the faster one's"
            if not getattr(args, "no_cpu_baseline", True) and "jpeg_crc32" in out and "cpu_baseline" not in out:
                # libjpeg-turbo on the box's host cores in the same run, at every N: the supervisor holds no GPU and its ranks are gone
                try:
                    out.update(cpu_baseline_fields(args, not args.no_optimize, out["config"]["restart_interval"], out["jpeg_crc32"], out["jpeg_bytes"]))
                except Exception as ex:      # noqa: BLE001 -- the measured line stands without its CPU legs
                    out["cpu_baseline_error"] = repr(ex)[:300]
            print(json.dumps(out), flush=True)
            return 0
        failures.append("%s: %s" % (mode, reason))
        print("[bench] gather '%s' failed (%s)%s" % (mode, reason, "; starting fresh ranks with the next one" if mode != ladder[-1] else ""),
              file=sys.stderr, flush=True)
    print("[bench] every gather failed: " + "; ".join(failures), file=sys.stderr, flush=True)
    return 1


def main():
    args = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    child = os.environ.get("MIJ_BENCH_CHILD") == "1" or os.environ.get("MIJ_BENCH_DIRECT") == "1"
    if not child and (env_world > 1 or args.gpus > 1):
        if env_world > 1:        # under an external launcher: its local rank 0 supervises, the others have nothing to do
            if int(os.environ.get("LOCAL_RANK", "0")) != 0:
                return 0
            return supervise(args, env_world, "external launcher (%d ranks): its local rank 0 supervised %d fresh child ranks" % (env_world, env_world))
        return supervise(args, args.gpus, "self: %d fresh child ranks started and watched by the bench process" % args.gpus)
    return worker(args)


def _progress(rank, what):
    """Rank 0 of a supervised run tells the supervisor how far it got (its watchdog is on the time between two marks)."""
    if rank == 0 and os.environ.get("MIJ_BENCH_CHILD") == "1":
        print("##progress " + what, flush=True)
        _HEART["t"] = time.monotonic()


_HEART = {"t": 0.0, "on": os.environ.get("MIJ_BENCH_CHILD") == "1"}


def _heartbeat(rank, what):
    """Inside loops: a mark at most every five seconds, so that a slow but healthy run (a rehearsal over gloo, a node with a slow
    gather) is not taken for a stalled one."""
    if _HEART["on"] and rank == 0 and time.monotonic() - _HEART["t"] > 5.0:
        _progress(rank, what)


def _inject(rank, gather, where):
    """Test hook (tests/test_gpu_sharded.py): MIJ_BENCH_INJECT="die|hang:<gather>:<rank>:<where>" makes that rank of a run with
    that gather exit or stall at that point, so that the supervisor's fallback can be exercised without breaking a GPU."""
    spec = os.environ.get("MIJ_BENCH_INJECT")
    if not spec:
        return
    kind, g, r, w = (spec.split(":") + ["", "", "", ""])[:4]
    if g == gather and int(r or 0) == rank and (w or "init") == where:
        if kind == "die":
            os._exit(3)
        time.sleep(10 ** 6)


def worker(args):
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return worker_one_gpu(args)
    return worker_ranks(args)


def _one_gpu_loop(args, optimize):
    """Which loop produces `value` on one GPU (see --loop)."""
    loop = args.loop
    if loop is None:
        if args.two_streams:
            loop = "two-streams"
        elif args.tables_ahead is False:
            loop = "one-stream"
        elif args.tables_ahead is True:
            loop = "tables-ahead"
        else:
            loop = "overlap"
    if args.progressive:
        return "progressive"          # one handle, the ten scans on the library's own four streams
    if loop == "tables-ahead" and not optimize:
        loop = "one-stream"           # fixed tables: nothing to build ahead
    return loop


def _clock(torch, mij, where):
    """Effective shader clock right now (mij_clock_probe_device: ~0.9 ms of a fixed vector-ALU loop on every SIMD)."""
    c = mij.clock_probe_device(1024, torch.cuda.current_stream().cuda_stream)
    return {"when": where, "valu_MHz": round(c["valu_mhz"], 1), "counter_MHz": round(c["counter_mhz"], 1)}


def _smi_clocks():
    """What the SMI tool says about clocks and the power cap, if an ordinary user may ask (informational; never fails the bench)."""
    import re
    import subprocess
    out = {}
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        # under rocprofv3 the profiler's preloaded library initialises the GPU in every process it is inherited by, and rocm-smi is a
        # script (env -> python3): that chain of exec()s from GPU-initialised processes is what the GPU boxes refuse
        return {"note": "not queried under a profiler"}
    try:
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
        r = subprocess.run(["rocm-smi", "--showclocks", "--showmaxpower", "--showpower", "--showperflevel"], capture_output=True, text=True, timeout=20, env=env)
        for key, pat in (("sclk_MHz", r"GPU\[0\].*sclk clock level.*\((\d+)Mhz\)"), ("mclk_MHz", r"GPU\[0\].*mclk clock level.*\((\d+)Mhz\)"),
                         ("power_cap_W", r"GPU\[0\].*Max Graphics Package Power \(W\):\s*([\d.]+)"),
                         ("socket_power_W", r"GPU\[0\].*Current Socket Graphics Package Power \(W\):\s*([\d.]+)"),
                         ("perf_level", r"GPU\[0\].*Performance Level:\s*(\w+)")):
            m = re.search(pat, r.stdout)
            if m:
                out[key] = m.group(1) if key == "perf_level" else float(m.group(1))
    except Exception as ex:      # noqa: BLE001
        out["error"] = repr(ex)[:120]
    out["note"] = "rocm-smi, read while the GPU is idle between the passes (sclk then shows a sleep state): informational"
    return out


def worker_one_gpu(args):
    import torch
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    optimize = not args.no_optimize
    W, H = args.width, args.height
    loop = _one_gpu_loop(args, optimize)
    n_handles = {"overlap": 3, "tables-ahead": 3, "one-stream": 2, "two-streams": 2, "progressive": 1}[loop]
    encs = [sharded.make_hip_strip_encoder(torch, W, H, args.quality, optimize, args.css, 0, 1, 0, args.fmt,
                                           restart_interval=args.restart_interval, progressive=args.progressive) for _ in range(n_handles)]
    geo = encs[0].geometry
    rows = geo["strip_rows"]
    main = torch.cuda.current_stream().cuda_stream
    d_img = torch.empty((rows, W, 3), dtype=torch.uint8, device=dev)
    mij.synth_image_device(d_img.data_ptr(), W, 0, rows, W * 3, bgr=(args.fmt == "bgr"), stream=main)
    strips = [sharded.HipStripEncoder(torch, e, d_img, args.fmt, shared_statistics=False) for e in encs]
    torch.cuda.synchronize()
    clocks = [_clock(torch, mij, "start (device idle before)")]
    copy_gbs, copy_lib_gbs = hbm_copy_ceiling(torch, mij, dev)

    # ---- the timed loop ---------------------------------------------------------------------------------------------------
    side = torch.cuda.Stream() if loop in ("overlap", "tables-ahead", "two-streams") else None
    side_s = side.cuda_stream if side is not None else main
    state = {"i": 0, "q": [], "last": None}            # q: issued and not yet collected, oldest first; last: the newest collected file

    def issue(st):
        e = st.enc
        if loop == "overlap":            # K1 on the main stream; K3 K4 K5 K6 on the side stream, behind the transform by event
            e.transform(st.d_img.data_ptr(), st.pitch, st.fmt, 0, main)
            if optimize:
                e.tables(side_s)
            e.entropy(side_s)
        elif loop == "tables-ahead":     # all wide kernels on the main stream; the one-workgroup table build on the side stream
            e.transform(st.d_img.data_ptr(), st.pitch, st.fmt, 0, main)
            e.tables(side_s)
            if state["q"]:
                state["q"][-1].enc.entropy(main)          # the previous image's entropy stage, behind this image's transform
        elif loop == "two-streams":
            st.issue_whole(main if state["i"] & 1 else side_s)
        else:                            # one-stream, progressive
            st.issue_whole(main)

    def drain():
        if loop == "tables-ahead" and state["q"]:
            state["q"][-1].enc.entropy(main)
        while state["q"]:
            state["last"] = state["q"].pop(0).finish_whole()
        return state["last"]

    def step():
        st = strips[state["i"] % n_handles]
        issue(st)
        state["i"] += 1
        state["q"].append(st)
        if len(state["q"]) >= n_handles:          # the handle the next step needs: collect its image while the others run
            state["last"] = state["q"].pop(0).finish_whole()
        return state["last"]

    def fence():
        torch.cuda.synchronize()

    # the stage-A pass's encoder (fixed tables) is made here: allocating its workspace between the passes below would idle the device
    # for long enough that the clock falls back (seen: 0.448 ms average over launches whose fastest took 0.414)
    ea = sa = ea2 = sa2 = None
    if args.stage_a_pass > 0 and not args.progressive and optimize:
        ea = sharded.make_hip_strip_encoder(torch, W, H, args.quality, False, args.css, 0, 1, 0, args.fmt, restart_interval=args.restart_interval)
        sa = sharded.HipStripEncoder(torch, ea, d_img, args.fmt, shared_statistics=False)
        ea2 = sharded.make_hip_strip_encoder(torch, W, H, args.quality, False, args.css, 0, 1, 0, args.fmt, restart_interval=args.restart_interval)
        sa2 = sharded.HipStripEncoder(torch, ea2, d_img, args.fmt, shared_statistics=False)
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    # The shader clock needs tens of milliseconds of THIS load to settle (from idle the first ~20 images of the loop run 5-25 % slower
    # while it ramps, tools/clock_transient.py): W warm-up steps of 1.1 ms each do not get there. So, still untimed, the same loop
    # runs on in rounds of ten steps until two successive clock readings agree within 1 % (at most `--settle-rounds` rounds).
    settle_steps = 0
    clocks.append(_clock(torch, mij, "after the warm-up steps"))
    for r in range(max(0, args.settle_rounds)):
        for _ in range(10):
            step()
        drain()
        fence()
        settle_steps += 10
        clocks.append(_clock(torch, mij, "after settle round %d" % (r + 1)))
        a, b = clocks[-2]["counter_MHz"] or clocks[-2]["valu_MHz"], clocks[-1]["counter_MHz"] or clocks[-1]["valu_MHz"]
        idle = clocks[0]["counter_MHz"] or clocks[0]["valu_MHz"]
        # settled = two successive readings agree AND the clock is not still sitting in the dip it takes when load arrives (below its
        # idle reading): on one box the dip lasted through the warm-up and the first settle round -- two equal LOW readings ended the
        # settling there and the 20 timed steps ran up the ramp (1.163 ms instead of 1.11-1.12)
        if abs(a - b) <= 0.01 * b and b >= 0.995 * idle:
            break
    clocks[-1]["when"] += " = right before the timed region"
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    jpeg_t = drain()
    fence()
    dt = time.perf_counter() - t0
    clocks.append(_clock(torch, mij, "right after the timed region"))
    ms_per_step = dt / args.steps * 1e3
    value = (W * H / 1e6) / (ms_per_step / 1e3)
    jpeg_keep = jpeg_t.clone()

    # ---- per-kernel pass: every kernel of an image alone on ONE stream, events around each (the library's own, mij_stage_times) ----
    # (Two handles take turns on the ONE stream: image i + 1 is enqueued before the host waits for image i, so the device never idles
    # between the images -- with one handle it did, ~50 us per image, and over a dozen images the clock started to fall: 2.42 -> 2.35 GHz
    # measured behind the stage-A pass -- while every kernel still runs alone, one behind the other on the same stream.)
    def alone_pass(pairs, count, warm, collect):
        n, total, out = len(pairs), count + warm, None
        for e, _ in pairs:
            e.enable_timing(True)
        pairs[0][1].issue_whole(main)
        issued = 1
        for i in range(total):
            if n > 1 and issued < total:
                pairs[issued % n][1].issue_whole(main)
                issued += 1
            e, st = pairs[i % n]
            out = st.finish_whole()
            if i >= warm:
                collect(e.stage_times())
            if n == 1 and issued < total:
                st.issue_whole(main)
                issued += 1
        for e, _ in pairs:
            e.enable_timing(False)
        return out

    stage_acc = {}
    kp = max(1, args.kernel_pass)

    def add_stages(t):
        for k, v in t.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    jpeg_kp = alone_pass(list(zip(encs, strips))[:2], kp, 2, add_stages)
    stages = {k: v / kp for k, v in stage_acc.items()}
    jpeg_kp = jpeg_kp.clone()            # fingerprinted in report(): a 204-MB copy to the host here would let the clock fall back
    clocks.append(_clock(torch, mij, "after the per-kernel pass"))

    # ---- stage A alone: the transform WITHOUT the fused statistics (what a fixed-table encoder runs), same pixels ----
    stage_a = None
    if args.stage_a_pass > 0 and not args.progressive:
        pairs_a = list(zip(encs, strips))[:2] if not optimize else [(ea, sa), (ea2, sa2)]
        each = []
        alone_pass(pairs_a, args.stage_a_pass, 4, lambda t: each.append(t["transform"]))
        acc = sum(each)
        ms_a = acc / args.stage_a_pass
        bytes_a = algorithmic_bytes_per_pixel(args.css, 0.0)["transform"] * rows * W
        stage_a = {"kernel": "k_transform without statistics (fixed-table encoder, same pixels)", "launches": args.stage_a_pass,
                   "avg_launch_ms": round(ms_a, 4), "min_launch_ms": round(min(each), 4), "achieved": round(bytes_a / (ms_a * 1e-3) / 1e9, 1),
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_a / (ms_a * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes_per_launch": int(bytes_a),
                   "note": "hipEvents around each launch on its stream, kernel alone on the device, after the timed region"}
        clocks.append(_clock(torch, mij, "after the stage-A pass"))
        if optimize:
            ea.close()
            ea2.close()
            del sa, ea, sa2, ea2

    # ---- one image alone: issue -> complete file (what the reference's one event pair around nvjpegEncodeImage times, .cu:279-291) ----
    lat = []
    for _ in range(5):
        fence()
        t1 = time.perf_counter()
        issue(strips[0])
        state["i"] += 1
        state["q"].append(strips[0])
        drain()
        lat.append((time.perf_counter() - t1) * 1e3)
    single_ms = sorted(lat)[len(lat) // 2]
    smi = _smi_clocks()
    for e in encs[1:]:
        e.close()

    ctx = dict(world=1, rank=0, W=W, H=H, optimize=optimize, geo=geo, rows=rows, value=value, ms_per_step=ms_per_step, jpeg_t=jpeg_keep,
               stages=stages, n_handles=n_handles, gather_mode=None, dpipe=None, copy_gbs=copy_gbs, copy_lib_gbs=copy_lib_gbs, d_img=d_img,
               root_files=None, want_put=False, one_device=False, rccl_ranks=None, single_ms=None, put_rate=None,
               streams=(1 if loop in ("one-stream", "progressive") else 2),
               pipeline={"overlap": "transform of image i on one stream, entropy stage (tables, coder, scan, compaction) of image i-1 on a second one; three handles",
                         "tables-ahead": "image i's table build (one workgroup) on a side stream under image i-1's entropy coder; all other kernels of all images on one stream",
                         "two-streams": "two handles, each image entirely on its own stream", "one-stream": "two handles, every kernel in issue order on one stream",
                         "progressive": "one handle; the ten scans on the library's four internal streams"}[loop])
    extra = {"loop": loop,
             "per_kernel_pass": {"images": kp, "streams": 1, "file_identical_to_timed_loop": None,
                                 "note": "`roofline`, `stage_roofline` and `stage_ms` come from this pass: one image at a time on ONE stream, every kernel "
                                         "alone on the device, right AFTER the timed region; the timed loop overlaps kernels of different images, so a "
                                         "kernel's duration inside it includes the sharing"},
             "single_image_ms": round(single_ms, 4),
             "single_image_note": "one image alone, issue to complete file with nothing else in flight (host clock, median of 5): the figure "
                                  "comparable with the reference's one event pair around nvjpegEncodeImage (ImageCompressorImpl.cu:279-291); its "
                                  "device-event counterpart is stage_ms.total",
             "clock": {"probes": clocks, "settle_steps": settle_steps, "smi": smi,
                       "note": "valu_MHz = issue rate of a fixed 4-cycle vector-ALU loop on every SIMD (8 waves per SIMD, ~0.9 ms, launch overhead "
                               "included, so a lower bound); counter_MHz = shader-clock counter over constant-rate counter inside the same launch"}}
    if stage_a is not None:
        extra["stage_A_alone"] = stage_a
    extra["_kp_file"] = jpeg_kp
    extra["untimed_steps"] = args.warmup + settle_steps
    extra["untimed_steps_note"] = "%d warm-up steps + %d settling steps of the same loop (clock.settle_steps) ran before the timed region" % (args.warmup, settle_steps)
    default_workload = ((W, H, args.quality, args.css) == (W_IMG, H_IMG, QUALITY, CSS_NAME) and optimize and not args.progressive
                        and args.restart_interval < 0)
    if default_workload and not args.no_extra:
        extra.update(extra_sections(args, torch, mij, sharded, d_img, jpeg_keep, copy_gbs))
    rc = report(args, torch, mij, ctx, extra)
    encs[0].close()
    return rc


# =====================================================================================================================
# Sections behind the headline passes (one GPU, default workload): BASELINE configs 3 and 5 and both decoders, so that the ONE
# driver-run line carries every single-GPU configuration. Each section is fenced, untimed by `value`, and fails soft (an "error" key).
# =====================================================================================================================
def _golden(name):
    try:
        with open(os.path.join(ROOT, "tests", "golden", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _psnr_device(torch, a, b, band=4000):
    """PSNR of two H x W x 3 uint8 device images (integer squared error summed in bands)."""
    import math
    se = 0
    for y in range(0, a.shape[0], band):
        d = a[y:y + band].to(torch.int32) - b[y:y + band].to(torch.int32)
        se += int((d * d).sum().item())
    n = a.numel()
    return float("inf") if se == 0 else round(10.0 * math.log10(255.0 ** 2 / (se / n)), 3)


def _crc_device(t):
    import numpy as np
    return "%08x" % zlib.crc32(np.ascontiguousarray(t.cpu().numpy()))


def _wall(torch, f, reps=3, warm=1):
    """Median wall ms of f() fenced by device synchronisation on both sides (what a caller with one image waits for)."""
    out = []
    for i in range(warm + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        if i >= warm:
            out.append((time.perf_counter() - t0) * 1e3)
    return round(sorted(out)[len(out) // 2], 4), [round(x, 4) for x in out]


def section_table1(args, torch, mij, sharded, d_img, copy_gbs):
    """BASELINE config 3 = the reference's table (1) (README.md:45-51): the five samplings at q95, optimised tables, same loop as the headline
    (transform of image i beside the entropy stage of image i-1, three handles), K1 alone from a one-stream pass with events."""
    W, H = W_IMG, H_IMG
    gold = _golden("big_8320x40000_q95.json").get("cases", {})
    main = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    side_s = side.cuda_stream
    steps, warm = 20, 10
    rows = {}
    out_img = torch.empty_like(d_img)
    with mij.Decoder() as dec:
        for css in ("444", "422", "440", "420", "411"):
            encs = [sharded.make_hip_strip_encoder(torch, W, H, QUALITY, True, css, 0, 1, 0, args.fmt) for _ in range(3)]
            strips = [sharded.HipStripEncoder(torch, e, d_img, args.fmt, shared_statistics=False) for e in encs]
            q = []

            def step(i):
                st = strips[i % 3]
                st.enc.transform(st.d_img.data_ptr(), st.pitch, st.fmt, 0, main)
                st.enc.tables(side_s)
                st.enc.entropy(side_s)
                q.append(st)
                return q.pop(0).finish_whole() if len(q) >= 3 else None

            def drain():
                last = None
                while q:
                    last = q.pop(0).finish_whole()
                return last
            for i in range(warm):
                step(i)
            drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            f = drain()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            jpg = f.clone()
            # K1 alone: two handles taking turns on one stream, events around every kernel (the library's own)
            acc, n_kp = {}, 6
            for e in encs[:2]:
                e.enable_timing(True)
            strips[0].issue_whole(main)
            for i in range(n_kp + 2):
                if i + 1 < n_kp + 2:
                    strips[(i + 1) % 2].issue_whole(main)
                strips[i % 2].finish_whole()
                if i >= 2:
                    for k, v in encs[i % 2].stage_times().items():
                        acc[k] = acc.get(k, 0.0) + v / n_kp
            # the file through the library's own decoder (pixel-identical to libjpeg-turbo's by the GPU tests; Pillow cannot be asked for
            # five 333-Mpixel decodes inside a bench line): PSNR against the source, decode time
            dms = []
            for _ in range(3):
                dec.decode_device_ptr(jpg.data_ptr(), jpg.numel(), out_img.data_ptr(), W * 3, args.fmt)
                dms.append(dec.sync())
            ri = encs[0].geometry["restart_interval"]
            for e in encs:
                e.close()
            nbytes = int(jpg.numel())
            crc = _crc_device(jpg)
            g = gold.get("css%d_ri%d_opt" % ({"444": 0, "422": 1, "420": 2, "440": 3, "411": 4}[css], ri))
            bpp = algorithmic_bytes_per_pixel(css, nbytes / (3.0 * W * H))
            k1 = bpp["transform"] * W * H / (acc["transform"] * 1e-3) / 1e9
            k4 = bpp["entropy"] * W * H / (acc["entropy"] * 1e-3) / 1e9
            rows[css] = {"ms_per_step": round(ms, 4), "Mpixels/s": round(W * H / 1e6 / (ms * 1e-3), 1), "steps": steps, "restart_interval": ri,
                         "jpeg_bytes": nbytes, "ratio": round(nbytes / (3.0 * W * H), 5), "crc32": crc,
                         "golden_match": (None if g is None else (g["crc32"] == crc and g["len"] == nbytes)),
                         "psnr_db": _psnr_device(torch, out_img, d_img),
                         "K1_ms": round(acc["transform"], 4), "K1_frac": round(k1 / HBM_PEAK_GBS, 4),
                         "K4_ms": round(acc["entropy"], 4), "K4_frac": round(k4 / HBM_PEAK_GBS, 4),
                         "stage_ms": {k: round(v, 4) for k, v in acc.items()},
                         "decode_ms": round(sorted(dms)[1], 3)}
            del strips, encs, jpg, f
    del out_img
    torch.cuda.empty_cache()
    return {"what": "BASELINE config 3 (reference README.md:45-51, table 1): 8320x40000 q95 optimised Huffman at the five samplings; %d timed steps each "
                    "of the headline's loop after %d warm-up steps; K1 / K4 alone from a one-stream pass of 6 images with hipEvents; golden = "
                    "tests/golden/big_8320x40000_q95.json (CPU oracle); psnr_db through the library's own decoder" % (steps, warm),
            "rows": rows}


def section_secondary(args, torch, mij, d_img):
    """BASELINE config 5: encode -> difference map (from the first layer's coefficients) -> encode; and back: decode both, add."""
    W, H = W_IMG, H_IMG
    gold = _golden("big_secondary_8320x40000.json")
    res_img, dec_img, rec_img = torch.empty_like(d_img), torch.empty_like(d_img), torch.empty_like(d_img)
    n = d_img.numel()
    out = {"what": "BASELINE config 5 (reference README.md:8): J1 = enc(I); R = clip((I - dec(J1)) * gain + 128) from J1's coefficients; J2 = enc(R); and "
                   "back: I' = clip(dec(J1) + (dec(J2) - 128) / gain). Wall ms fenced by device synchronisation, median of 3 after one warm-up, "
                   "everything device resident; golden = tests/golden/big_secondary_8320x40000.json (CPU oracle + libjpeg-turbo's decode)",
           "first_layer": None, "cases": {}}
    with mij.Encoder(W, H, QUALITY, True, 1) as e1, mij.Decoder() as dec:
        r1 = {}
        for key, q2, css2, gain in (("q95_css1_gain1", 95, 1, 1), ("q98_css0_gain1", 98, 0, 1)):
            with mij.Encoder(W, H, q2, True, css2) as e2:
                r2 = {}

                def compress():
                    e1.encode_device(d_img.data_ptr(), W * 3, args.fmt)
                    r1.update(e1.result())
                    e1.residual_device(d_img.data_ptr(), W * 3, res_img.data_ptr(), args.fmt, gain=gain)
                    e2.encode_device(res_img.data_ptr(), W * 3, args.fmt)
                    r2.update(e2.result())

                def expand():
                    # (the two layers decoded side by side on two handles -- a host thread and a stream each -- measured 9.1 ms against 8.6 in
                    #  sequence for ONE pair: the overlap pays in a steady stream of files, decode.own_file.two_in_flight, not in a single pair)
                    dec.decode_device_ptr(r1["d_buffer"] + r1["header_offset"], r1["file_bytes"], dec_img.data_ptr(), W * 3, args.fmt)
                    dec.sync()
                    dec.decode_device_ptr(r2["d_buffer"] + r2["header_offset"], r2["file_bytes"], rec_img.data_ptr(), W * 3, args.fmt)
                    dec.sync()
                    mij.residual_device(dec_img.data_ptr(), rec_img.data_ptr(), rec_img.data_ptr(), n, +1, gain=gain)
                c_ms, c_all = _wall(torch, compress)
                x_ms, x_all = _wall(torch, expand)
                j2 = e2.retrieve()
                g = gold.get("cases", {}).get(key)
                crc2 = "%08x" % zlib.crc32(j2)
                case = {"quality2": q2, "css2": {0: "444", 1: "422"}[css2], "gain": gain, "secondary_compress_ms": c_ms, "runs_ms": c_all,
                        "decode_both_and_add_ms": x_ms, "decode_runs_ms": x_all, "round_trip_ms": round(c_ms + x_ms, 4),
                        "bytes": [r1["file_bytes"], len(j2)], "crc32_second_layer": crc2,
                        "golden_match": (None if g is None else (g["crc32"] == crc2 and g["len"] == len(j2))),
                        "psnr_first_layer_db": _psnr_device(torch, dec_img, d_img), "psnr_both_layers_db": _psnr_device(torch, rec_img, d_img)}
                if g is not None:
                    case["golden_psnr_db"] = [g.get("psnr_first_layer"), g.get("psnr_both_layers")]
                out["cases"][key] = case
        j1 = e1.retrieve()
        g1 = gold.get("first_layer")
        crc1 = "%08x" % zlib.crc32(j1)
        out["first_layer"] = {"bytes": len(j1), "crc32": crc1, "golden_match": (None if not g1 else (g1["crc32"] == crc1 and g1["len"] == len(j1)))}
    del res_img, dec_img, rec_img
    torch.cuda.empty_cache()
    return out


def section_decode(args, torch, mij, d_img, d_file, copy_gbs):
    """A10: the library's decoder on (i) its own full-size headline file, (ii) the reference's format -- libjpeg-turbo's progressive file of the
    same image without restart markers, written by Pillow on the host in this run."""
    W, H = W_IMG, H_IMG
    out = {}
    d_out = torch.empty_like(d_img)
    nfile = int(d_file.numel())
    with mij.Decoder() as dec:
        ms = []
        for _ in range(7):
            dec.decode_device_ptr(d_file.data_ptr(), nfile, d_out.data_ptr(), W * 3, "rgb")
            ms.append(dec.sync())
        ms = ms[2:]
        med = sorted(ms)[len(ms) // 2]
        alg = nfile + 3 * W * H
        gbs = alg / (med * 1e-3) / 1e9
        out["own_file"] = {"file": "the timed region's file (baseline, DRI=64), device resident in and out", "file_bytes": nfile,
                           "device_ms": round(med, 3), "runs_ms": [round(x, 3) for x in ms], "Mpixels/s": round(W * H / 1e6 / (med * 1e-3), 1),
                           "decoded_crc32": _crc_device(d_out), "identical_to_pillow": None,
                           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                        "frac_of_copy_ceiling": round(gbs / copy_gbs, 4) if copy_gbs else None,
                                        "algorithmic_bytes": alg, "note": "algorithmic bytes = file + 3*W*H pixels written; device time of the whole decode "
                                        "(mij_decode_sync: events around all its kernels), median of 5 after 2 warm-up decodes. Huffman decoding is "
                                        "bound by instruction issue / serial dependence, not by HBM (DESIGN.md section 6)"}}
        # throughput with two files in flight (two handles, one host thread and one stream each): the dense passes of one decode are a single round
        # of 3.7 waves per SIMD, so a second file fills what the first leaves idle
        try:
            import threading
            with mij.Decoder() as dec2:
                outs = [d_out, torch.empty_like(d_out)]
                sts = [torch.cuda.Stream(), torch.cuda.Stream()]
                decs, reps = [dec, dec2], 8

                def work(k):
                    for _ in range(reps):
                        decs[k].decode_device_ptr(d_file.data_ptr(), nfile, outs[k].data_ptr(), W * 3, "rgb", 0, sts[k].cuda_stream)
                        decs[k].sync()
                for k in range(2):
                    decs[k].decode_device_ptr(d_file.data_ptr(), nfile, outs[k].data_ptr(), W * 3, "rgb", 0, sts[k].cuda_stream)
                    decs[k].sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
                [t.start() for t in th]
                [t.join() for t in th]
                torch.cuda.synchronize()
                out["own_file"]["two_in_flight"] = {"ms_per_file": round((time.perf_counter() - t0) / (2 * reps) * 1e3, 3), "files": 2 * reps,
                                                    "identical_outputs": bool(torch.equal(outs[0], outs[1])),
                                                    "note": "wall time per file with two decoder handles in flight, one host thread and stream each"}
                del outs
        except Exception as ex:      # noqa: BLE001
            out["own_file"]["two_in_flight"] = {"error": repr(ex)[:200]}
        if not args.no_progressive_decode:
            from PIL import Image, ImageFile
            Image.MAX_IMAGE_PIXELS = None
            ImageFile.MAXBLOCK = 1 << 30
            gp = _golden("prog_nodri_8320x40000.json")
            img = d_img.cpu().numpy()
            if args.fmt == "bgr":
                img = img[..., ::-1]
            t0 = time.perf_counter()
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", quality=QUALITY, subsampling=1, progressive=True, optimize=True)
            j = b.getvalue()
            t_enc = time.perf_counter() - t0
            del img, b
            d_j = torch.frombuffer(bytearray(j), dtype=torch.uint8).cuda()
            pms = []
            for _ in range(4):
                dec.decode_device_ptr(d_j.data_ptr(), len(j), d_out.data_ptr(), W * 3, "rgb")
                pms.append(dec.sync())
            tried, par = dec.px_report()
            pmed = sorted(pms[1:])[1]
            crc_f, crc_d = "%08x" % zlib.crc32(j), _crc_device(d_out)
            out["progressive_nodri"] = {"file": "libjpeg-turbo (Pillow) progressive SOF2, optimised tables, NO restart markers, q95 4:2:2 of the same image "
                                                "(the reference's nvJPEG output mode, ImageCompressorImpl.cu:28), written on the host in this run",
                                        "file_bytes": len(j), "file_crc32": crc_f, "pillow_encode_s": round(t_enc, 1),
                                        "device_ms": round(pmed, 2), "runs_ms": [round(x, 2) for x in pms], "scans_tried": tried, "scans_parallel": par,
                                        "decoded_crc32": crc_d,
                                        "golden": {"file_crc32": gp.get("file_crc32"), "decoded_crc32": gp.get("decoded_crc32"),
                                                   "source": "tests/golden/prog_nodri_8320x40000.json: Pillow's own decode of that file"},
                                        "golden_match": (None if not gp else (gp.get("file_crc32") == crc_f and gp.get("decoded_crc32") == crc_d)),
                                        "Mpixels/s": round(W * H / 1e6 / (pmed * 1e-3), 1)}
            del d_j
    del d_out
    torch.cuda.empty_cache()
    return out


def extra_sections(args, torch, mij, sharded, d_img, d_file, copy_gbs):
    out, t_all = {"configs": {}}, time.perf_counter()
    for name, fn in (("table1", lambda: section_table1(args, torch, mij, sharded, d_img, copy_gbs)),
                     ("secondary", lambda: section_secondary(args, torch, mij, d_img))):
        t0 = time.perf_counter()
        try:
            out["configs"][name] = fn()
        except Exception as ex:      # noqa: BLE001 -- the headline stands without a section that failed; the failure is in the line
            out["configs"][name] = {"error": repr(ex)[:400]}
        out["configs"][name]["section_wall_s"] = round(time.perf_counter() - t0, 1)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        out["decode"] = section_decode(args, torch, mij, d_img, d_file, copy_gbs)
    except Exception as ex:          # noqa: BLE001
        out["decode"] = {"error": repr(ex)[:400]}
    out["decode"]["section_wall_s"] = round(time.perf_counter() - t0, 1)
    out["extra_sections_wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


def worker_ranks(args):
    import datetime
    import torch
    import torch.distributed as dist
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _progress(rank, "imported")
    if args.gather == "auto":          # (a rank started directly, without the supervisor)
        args.gather = "put"
    if args.gather == "serial":
        os.environ["MIJ_BENCH_NO_PIPELINE"] = "1"
    # Rehearsal switches (tests only): all ranks on GPU 0 with gloo carrying the collectives, because RCCL refuses two ranks
    # on one device. The measured runs use one GPU per rank over RCCL/xGMI.
    one_device = os.environ.get("MIJ_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rccl_ranks = None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # A collective that does not complete within the bound ends the rank (RCCL's watchdog), the supervisor sees it die
    # and starts fresh processes with the next gather: nothing waits for ever.
    coll_timeout = datetime.timedelta(seconds=float(os.environ.get("MIJ_BENCH_COLL_TIMEOUT_S", "120")))
    if one_device:
        dist.init_process_group("gloo", timeout=coll_timeout)
    else:
        dist.init_process_group("nccl", device_id=dev, timeout=coll_timeout)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)                    # how many ranks RCCL really joined
        rccl_ranks = int(ones.item())
    _progress(rank, "process group up")
    _inject(rank, args.gather, "init")
    optimize = not args.no_optimize
    W, H = args.width, args.height

    # ---- strip partition (pure arithmetic, no communication) + this rank's pixels --------------------------------
    ri_arg = args.restart_interval
    if args.progressive and ri_arg < 0:       # sharded progressive: the interval must divide the MCU row (sharded.progressive_strip_interval)
        ri_arg = sharded.progressive_strip_interval(sharded.strip_rows(W, H, args.quality, optimize, args.css, 0, 1, -1)[0]["mcus_per_row"])
    whole_geo, r0_, r1_ = sharded.strip_rows(W, H, args.quality, optimize, args.css, rank, world, ri_arg)
    want_put = args.gather == "put" and os.environ.get("MIJ_BENCH_NO_PIPELINE") != "1" and not args.progressive
    n_handles = 1 if args.progressive or os.environ.get("MIJ_BENCH_NO_PIPELINE") == "1" else (sharded.DEPTH if want_put else 2)
    encs = [sharded.make_hip_strip_encoder(torch, W, H, args.quality, optimize, args.css, rank, world, dev_index, args.fmt,
                                           restart_interval=args.restart_interval, progressive=args.progressive) for _ in range(n_handles)]
    enc = encs[0]                     # None: this rank owns no strip (more ranks than restart-aligned strips)
    geo = enc.geometry if enc is not None else dict(whole_geo, strip_y0=0, strip_rows=0)
    y0, rows = geo["strip_y0"], geo["strip_rows"]
    stream = torch.cuda.current_stream().cuda_stream
    d_img, strips = None, [None] * n_handles
    if enc is not None:
        d_img = torch.empty((rows, W, 3), dtype=torch.uint8, device=dev)
        mij.synth_image_device(d_img.data_ptr(), W, y0, rows, W * 3, bgr=(args.fmt == "bgr"), stream=stream)
        if args.progressive:
            strips = [sharded.HipProgressiveStrip(torch, e, d_img, args.fmt) for e in encs]
        else:
            strips = [sharded.HipStripEncoder(torch, e, d_img, args.fmt, shared_statistics=True) for e in encs]
    strip = strips[0]
    torch.cuda.synchronize()

    def one_image():
        """One image at a time, host-synchronised (the serial gather; the only form progressive output has)."""
        st = torch.cuda.current_stream().cuda_stream
        if args.progressive:
            return sharded.encode_step_progressive(torch, dist, strip, cache, st)
        return sharded.encode_step(torch, dist, strip, optimize, cache, st)

    copy_gbs, copy_lib_gbs = hbm_copy_ceiling(torch, mij, dev) if rank == 0 else (None, None)
    cache, stage_acc = {"device": dev}, {}

    # Images in flight. Every step still produces a complete file inside the timed region.
    #  * "put" (sharded.DevicePipeline): four images in flight, sizes all-gathered device to device, strips written straight into the
    #    peer-mapped buffer of the image's root (which rotates over the ranks); no host wait inside a step. If the buffers cannot be
    #    mapped, or with --gather sendrecv: sharded.StripPipeline (two in flight, sizes via the host, RCCL send/recv).
    #    MIJ_BENCH_NO_PIPELINE=1: one image at a time (sharded.encode_step).
    pipelined = n_handles > 1
    gather_mode, pipe, dpipe = "sendrecv", None, None
    if want_put:
        targets = sharded.open_file_targets(torch, dist, strips if enc is not None else None, rank, world, dev_index, whole_geo)
        if targets is not None:
            dpipe = sharded.DevicePipeline(torch, dist, strips if enc is not None else None, targets, optimize, device=dev,
                                           rotate=not args.fixed_root, comms=args.comms)
            gather_mode = "put"
    if dpipe is None and pipelined:
        unit = sharded.rows_per_restart_unit(whole_geo["mcus_per_row"], whole_geo["restart_interval"])
        if (whole_geo["mcu_rows"] + unit - 1) // unit < world:      # the same arithmetic on every rank: all of them stop
            raise SystemExit("the send/recv pipeline needs a strip on every rank (more ranks than restart-aligned strips)")
        pipe = sharded.StripPipeline(torch, dist, strips[:2], optimize)
    _progress(rank, "pipeline ready (%s)" % gather_mode)
    _inject(rank, args.gather, "pipeline")
    timed_handles = [] if dpipe is not None else [e for e in encs if e is not None]
    for e in timed_handles:
        e.enable_timing(True)

    def record_times(e):
        for k, v in e.stage_times().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v

    def collect(record):
        if dpipe is not None:                     # completes every image in flight; rank 0 gets the last file
            return dpipe.flush()
        if pipe is not None:                      # the last image of a multi-rank run
            prev = pipe.pending
            out = pipe.flush()
            if record and prev is not None:
                record_times(prev[0].enc)
            return out
        return None

    def step(record):
        if not pipelined:
            out = one_image()
            if record and not args.progressive:
                record_times(enc)
            return out
        if dpipe is not None:
            dpipe.step()                          # enqueue only: no host wait, no result yet
            return None
        prev = pipe.pending
        out = pipe.step()                         # issues this image, then completes the previous one
        if record and prev is not None:
            record_times(prev[0].enc)
        return out

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(False)
        _heartbeat(rank, "warm-up step %d" % i)
    collect(False)

    def verify_roots(what):
        """One image per ROOT collected and fingerprinted (the timed loop only ever looks at its last file). Every root must have
        assembled the same bytes; rank 0 compares them with the timed region's file."""
        mine = []
        for i in range(dpipe.nroots):
            _heartbeat(rank, "%s: verifying root %d" % (what, i))
            dpipe.step()
            o = dpipe.collect()
            if o is not None:
                b = o.cpu().numpy().tobytes()
                mine.append("%08x:%d" % (zlib.crc32(b), len(b)))
        every = [None] * world
        dist.all_gather_object(every, mine)
        files = [c for lst in every for c in lst]
        if len(set(files)) != 1 or len(files) != dpipe.nroots:
            raise SystemExit("put pipeline (%s): the %d roots assembled different files: %s" % (what, dpipe.nroots, files))
        return files

    root_files = verify_roots("before the timed region") if dpipe is not None else None
    fence()
    _progress(rank, "warm-up done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        jpeg_t = step(True)
        _heartbeat(rank, "timed step %d" % i)
    if pipelined:
        jpeg_t = collect(True)         # the last image of the timed region
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = (W * H / 1e6) / (ms_per_step / 1e3)
    steps_timed = args.steps
    _progress(rank, "timed region done")
    jpeg_keep = jpeg_t.clone() if jpeg_t is not None else jpeg_t     # later images reuse the buffers
    single_ms, put_rate = None, None
    if dpipe is not None and dpipe.last_root != 0:
        # the last file was assembled on another rank: bring it to rank 0 for the checks below (untimed)
        lr = dpipe.last_root
        nb = torch.zeros(1, dtype=torch.int64, device=dev)
        if rank == lr:
            nb[0] = jpeg_keep.numel()
        dist.broadcast(nb, src=lr)
        if rank == lr:
            dist.send(jpeg_keep, dst=0)
        elif rank == 0:
            jpeg_keep = torch.empty(int(nb.item()), dtype=torch.uint8, device=dev)
            dist.recv(jpeg_keep, src=lr)
    if dpipe is not None:
        # the same per-root check on images assembled AFTER the timed region (buffers, mappings and streams as the loop left them)
        root_files = root_files + verify_roots("after the timed region")
    # ONE image alone through the same path, start to complete file (what a caller with a single image waits for; the
    # `value` above is the rate with several images in flight). Max over ranks, median of five.
    lat = []
    for i in range(5):
        _heartbeat(rank, "latency %d" % i)
        fence()
        t1 = time.perf_counter()
        if dpipe is not None:
            dpipe.step()
            dpipe.flush()
        elif pipe is not None:
            pipe.step()
            pipe.flush()
        else:
            one_image()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    tl = torch.tensor(lat, dtype=torch.float64, device=dev)
    dist.all_reduce(tl, op=dist.ReduceOp.MAX)
    single_ms = sorted(tl.tolist())[len(lat) // 2] * 1e3
    if dpipe is not None:
        # what a link gives k_put: strip bytes / event time around the put, one image alone (no other traffic), a few roots
        samples = []
        for e in encs:
            if e is not None:
                e.enable_timing(True)
        for _ in range(max(1, min(3, dpipe.nroots))):
            fence()
            k = dpipe.step()
            root = dpipe.roots[k]
            dpipe.flush()
            if enc is not None and rank != root:
                _, p_ms = encs[k].place_times()
                samples.append((rank, root, int(dpipe.sizes[k][rank].item()), p_ms))
        for e in encs:
            if e is not None:
                e.enable_timing(False)
        every = [None] * world
        dist.all_gather_object(every, samples)
        rates = sorted(b / (ms * 1e-3) / 1e9 for lst in every for (_, _, b, ms) in lst if ms > 0 and b > 0)
        if rates:
            put_rate = {"min": round(rates[0], 2), "median": round(rates[len(rates) // 2], 2), "max": round(rates[-1], 2), "samples": len(rates),
                        "strip_MB": round(max(b for lst in every for (_, _, b, _) in lst) / 1e6, 2),
                        "note": "k_put alone on its stream, hipEvents around it; one image in flight"}
    _progress(rank, "latency and put rate done")
    if dpipe is not None:
        # The put pipeline records no per-stage events (nothing in it touches the host). For the stage table, code one more
        # image the host-synchronised way, outside the timed region, with events on.
        try:
            if enc is not None:
                enc.enable_timing(True)
            sharded.encode_step(torch, dist, strip, optimize, cache, torch.cuda.current_stream().cuda_stream)
            if enc is not None:
                record_times(enc)
            steps_timed = 1
        except Exception as ex:      # noqa: BLE001 -- the stage table is informational; the measured value stands without it
            print("stage-time pass failed: %r" % (ex,), file=sys.stderr)
            stage_acc.clear()
        fence()
    # so that a first multi-GPU run explains itself: every rank's stage table (the slowest one's is quoted) and what the root takes in per image
    mine = {k: v / steps_timed for k, v in stage_acc.items()}
    every_stage = [None] * world
    dist.all_gather_object(every_stage, {"rank": rank, "strip_rows": rows, "stage_ms": {k: round(v, 4) for k, v in mine.items()}})
    inbound = None
    if dpipe is not None:
        sz = [int(x) for x in dpipe.sizes[0].tolist()]              # strip bytes of the last image assembled in slot 0, all ranks (device-side all-gather)
        root0 = dpipe.roots[0]
        inbound = {"root": root0, "strip_bytes_per_rank": sz, "inbound_bytes_per_image": int(sum(sz) - sz[root0]),
                   "note": "what the other ranks put into the root's peer-mapped buffer for one image (the root's own strip is placed locally); the root "
                           "rotates over the ranks, so per GPU and image on average 1/%d of this arrives" % max(1, dpipe.nroots)}
    rc = 0
    if rank == 0:
        ctx = dict(world=world, rank=rank, W=W, H=H, optimize=optimize, geo=geo, rows=rows, value=value, ms_per_step=ms_per_step, jpeg_t=jpeg_keep,
                   stages={k: v / steps_timed for k, v in stage_acc.items()}, n_handles=n_handles, gather_mode=gather_mode, dpipe=dpipe,
                   copy_gbs=copy_gbs, copy_lib_gbs=copy_lib_gbs, d_img=None, root_files=root_files, want_put=want_put, one_device=one_device,
                   rccl_ranks=rccl_ranks, single_ms=single_ms, put_rate=put_rate, streams=n_handles if dpipe is not None else 1, pipeline=None)
        with_t = [e for e in every_stage if e and e["stage_ms"].get("total", 0) > 0]
        slow = max(with_t, key=lambda e: e["stage_ms"]["total"]) if with_t else None
        extra_n = {"slowest_rank": slow, "per_rank_total_ms": {str(e["rank"]): e["stage_ms"].get("total") for e in every_stage if e},
                   "root_inbound": inbound}
        rc = report(args, torch, mij, ctx, extra_n)
    for e in encs:
        if e is not None:
            e.close()
    dist.barrier()
    dist.destroy_process_group()
    return rc


def report(args, torch, mij, c, extra):
    """Rank 0: verify the file, build and print the JSON line."""
    world, W, H, optimize, geo, rows = c["world"], c["W"], c["H"], c["optimize"], c["geo"], c["rows"]
    dpipe, root_files, stages, copy_gbs, copy_lib_gbs = c["dpipe"], c["root_files"], c["stages"], c["copy_gbs"], c["copy_lib_gbs"]
    jpeg = c["jpeg_t"].cpu().numpy().tobytes()
    fingerprint = "%08x:%d" % (zlib.crc32(jpeg), len(jpeg))
    if root_files is not None and set(root_files) != {fingerprint}:
        raise SystemExit("put pipeline: the files collected from the roots (%s) differ from the timed region's file (%s)" % (root_files, fingerprint))
    kp_file = extra.pop("_kp_file", None)
    if kp_file is not None:
        same = kp_file.cpu().numpy().tobytes() == jpeg
        extra["per_kernel_pass"]["file_identical_to_timed_loop"] = same
        if not same:
            raise SystemExit("the one-stream pass and the timed loop produced different files")
    ratio = len(jpeg) / (3.0 * W * H)
    strip_px = rows * W
    bpp = algorithmic_bytes_per_pixel(args.css, ratio)
    kname = {"transform": "k_transform", "entropy": "k_encode", "compact": "k_compact"}
    stage_roof = {}
    for k in kname:
        if args.progressive and k != "transform":
            continue          # progressive: the ten scans are reported together under stage_ms["tables"]
        if stages.get(k, 0) > 0:
            gbs = bpp[k] * strip_px / (stages[k] * 1e-3) / 1e9
            stage_roof[k] = {"kernel": kname[k], "ms": round(stages[k], 4), "GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                             "frac_of_copy_ceiling": round(gbs / copy_gbs, 4) if copy_gbs else None,
                             "algorithmic_bytes_per_launch": int(bpp[k] * strip_px)}
    lib_hash = mij.library_source_hash()
    clock_mhz = None
    probes = extra.get("clock", {}).get("probes") or []
    around = [p["counter_MHz"] or p["valu_MHz"] for p in probes if "timed region" in p["when"]]
    if around:
        clock_mhz = sum(around) / len(around)
    if stage_roof and not args.progressive:
        dom = max(stage_roof, key=lambda k: stage_roof[k]["ms"])
        traffic, traffic_src = measured_traffic(kname[dom], args, optimize and not args.progressive, world, lib_hash)
        roofline = {"bound": "hbm", "kernel": kname[dom], "achieved": stage_roof[dom]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": stage_roof[dom]["frac"], "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": stage_roof[dom]["algorithmic_bytes_per_launch"], "avg_launch_ms": stage_roof[dom]["ms"],
                    "note": ("avg launch duration from hipEvents around the kernel on its stream, %s"
                             % ("in the separate one-stream pass after the timed region (per_kernel_pass)" if world == 1 else "over the timed steps"))}
        if _VALU.get(kname[dom]):
            # What bounds this path: vector instruction issue. A BOUND, not a floor of 4 cycles per instruction (rounds 3-4: K4
            # already ran under that one): the dynamic count is SQ_INSTS_VALU (PMC pass, same library hash), the cost per instruction the
            # class mix of the kernel's ISA -- 2 cycles for v_add/sub_u32, logic, v_ashrrev_i32, moves and scalar-width f32 add / mul / fma, 4 for the
            # rest (tools/valu_bound.py, rates measured by tools/valu_rate.hip) -- on 1,024 SIMDs at the clock measured around the timed region.
            n = _VALU[kname[dom]]
            mhz = clock_mhz or 2400.0
            classes = valu_classes(lib_hash)
            short = {v: k for k, v in kname.items()}
            per, bound_all, flat_all = {}, 0.0, 0.0
            for kn, cnt in n.items():
                cpi = (classes.get(kn) or {}).get("cycles_per_vector_instruction_bound")
                st = short.get(kn)
                ms_k = stages.get(st) if st else None
                e = {"wave_instructions": cnt}
                if cpi:
                    e["bound_cycles_per_instruction"] = cpi
                    e["bound_ms"] = round(cnt * cpi / (1024 * mhz * 1e6) * 1e3, 4)
                    e["share_4_cycle"] = classes[kn].get("share_4_cycle")
                    bound_all += e["bound_ms"]
                if ms_k:
                    e["launch_ms"] = round(ms_k, 4)
                    e["cycles_per_valu_instruction"] = round(ms_k * 1e-3 * 1024 * mhz * 1e6 / cnt, 3)       # measured: launch time x SIMDs x clock / count
                    if cpi:
                        e["bound_frac_of_launch"] = round(e["bound_ms"] / ms_k, 3)
                flat_all += cnt * 4.0 / (1024 * mhz * 1e6) * 1e3
                per[kn] = e
            roofline["valu_issue"] = {"model": "bound" if classes else "flat 4 cycles per instruction (no class file for this build: an upper estimate, not a bound)",
                                      "clock_MHz": round(mhz, 1),
                                      "clock_source": "measured around the timed region (clock.probes)" if clock_mhz else "assumed",
                                      "kernels": per, "all_kernels_bound_ms": round(bound_all, 4) if classes else None,
                                      "all_kernels_at_4_cycles_ms": round(flat_all, 4),
                                      "classes_source": classes.get("_source"),
                                      "note": "bound_ms = SQ_INSTS_VALU x (2 x share_2 + 4 x share_4 + 8 x share_8) / (1,024 SIMDs x clock); cycles_per_valu_instruction = "
                                              "what the launch actually spent per vector instruction. Their ratio is what is left to an ideal schedule of the "
                                              "SAME instructions (every 2-cycle slot filled by a second wave); fewer instructions move the bound itself"}
        if dom == "transform" and optimize:
            # K1 also takes the AC statistics (SURVEY 8d stage B, a separate 2(1+f) B/px read in an unfused design);
            # against stage A + B's algorithmic bytes, as SURVEY 8d prescribes for a fused kernel:
            fused = (bpp["transform"] + bpp_stage_b(args.css)) * strip_px / (stage_roof[dom]["ms"] * 1e-3) / 1e9
            roofline["frac_vs_stage_A_plus_B_bytes"] = round(fused / HBM_PEAK_GBS, 4)
        sa = extra.pop("stage_A_alone", None)
        if sa is not None:
            t2, src2 = measured_traffic("k_transform_nostats", args, True, world, lib_hash)
            sa["traffic"], sa["traffic_source"] = t2, src2
            sa["frac_of_copy_ceiling"] = round(sa["achieved"] / copy_gbs, 4) if copy_gbs else None
            roofline["stage_A_alone"] = sa
    else:   # --progressive: twenty lane-per-interval passes sequenced by the host; not a roofline-shaped workload
        roofline = {"bound": "hbm", "kernel": "k_prog_encode", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        extra.pop("stage_A_alone", None)
    out = {
        "metric": "Mpixels/s encode (+PSNR, ratio) 8320x40000 q95 4:2:2 @1/2/4/8 GPU",
        "value": round(c["value"], 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(c["ms_per_step"], 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%dx%d RGB8 (%s interleaved, device resident) -> %s JFIF, q%d, 4:%s:%s, %s Huffman, "
                               "DRI=%d MCUs" % (W, H, args.fmt.upper(), "progressive (SOF2)" if args.progressive else "baseline", args.quality,
                                                args.css[1], args.css[2], "optimised" if optimize else "fixed", geo["restart_interval"]),
                   "parallelism": "strips%d" % world, "restart_interval": geo["restart_interval"],
                   "images_in_flight": c["n_handles"], "gather": c["gather_mode"] if os.environ.get("MIJ_BENCH_NO_PIPELINE") != "1" or world == 1 else "serial",
                   "comms": (dpipe.comms if dpipe is not None else None),
                   "root": (None if dpipe is None else ("rank 0" if args.fixed_root else "rotating over the ranks, image by image")),
                   "streams": c["streams"], "pipeline": c["pipeline"]},
        "jpeg_bytes": len(jpeg), "ratio": round(ratio, 5), "jpeg_crc32": "%08x" % zlib.crc32(jpeg),
        "roofline": roofline, "hbm_copy_ceiling_GB/s": round(copy_gbs, 1) if copy_gbs else None,
        "hbm_copy_ceiling_note": "own 16-B/lane copy kernel (k_copy16), read + write bytes; torch copy_ on the same box: %s GB/s" % (round(copy_lib_gbs, 1) if copy_lib_gbs else None), "stage_roofline": stage_roof, "stage_ms": {k: round(v, 4) for k, v in stages.items()},
    }
    out.update(extra)
    out["library_source_hash"] = lib_hash
    golden = golden_fingerprint(args, optimize, geo["restart_interval"])
    if golden is not None:          # the committed fingerprint of this exact configuration (tests/golden/, computed on the CPU)
        out["golden_match"] = golden == (out["jpeg_crc32"], out["jpeg_bytes"])
    if world > 1:
        out["rccl_ranks"] = c["rccl_ranks"]
        out["collective_backend"] = "gloo (one-device rehearsal)" if c["one_device"] else "nccl (RCCL)"
        out["single_image_latency_ms"] = round(c["single_ms"], 4) if c["single_ms"] is not None else None
        out["single_image_latency_note"] = "one image alone, issue to complete file on its root, max over ranks, median of 5"
        out["put_GB/s"] = c["put_rate"]
        if root_files is not None:
            out["files_verified"] = {"roots": dpipe.nroots, "images": len(root_files), "identical_to_timed_file": True,
                                     "when": "one image per root before the timed region and one per root after it"}
        if c["want_put"] and dpipe is None:
            out["gather_note"] = "the output buffers could not be peer-mapped (hipIpc*): every rank took send/recv"
    if not args.no_psnr:
        out["psnr_db"], out["psnr_note"], pillow_crc = _psnr_check(jpeg, W, H, args.fmt, c["d_img"] if world == 1 else None)
        own = out.get("decode", {}).get("own_file")
        if own is not None:        # the library's decode of the timed region's file against libjpeg-turbo's decode of the same file, by CRC of the RGB8 image
            own["pillow_decoded_crc32"] = pillow_crc
            own["identical_to_pillow"] = own["decoded_crc32"] == pillow_crc
    if world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_fields(args, optimize, geo["restart_interval"], out["jpeg_crc32"], out["jpeg_bytes"]))
    print(json.dumps(out), flush=True)
    return 0


def cpu_baseline_fields(args, optimize, restart_interval, jpeg_crc32, jpeg_bytes):
    """The CPU legs of the line (run by rank 0 on one GPU, by the supervisor -- which holds no GPU -- at N > 1)."""
    cb = cpu_baselines(args, optimize, restart_interval)
    out = {}
    # The baseline north_star names is libjpeg-turbo on the box's host cores; kind "port" = the stock CPU library whose
    # arithmetic oracle/ restates byte for byte (there is no oracle/_ref: the reference's nvJPEG cannot be built here).
    if "turbo" in cb:
        out["cpu_baseline"] = dict(cb["turbo"], kind="port", impl="libjpeg-turbo 3.1.4.1 via Pillow (SIMD), all host cores")
        one = dict(cb["turbo_1core"], impl="libjpeg-turbo 3.1.4.1 via Pillow, one core (the library's native mode)")
        out["cpu_baseline_1core"] = one
        if one.get("crc32") is not None and one.get("whole_image"):
            # libjpeg-turbo wrote the WHOLE image on the host in this same run: its file against the GPU's, byte for byte by fingerprint
            out["turbo_file_identical"] = (one["crc32"] == jpeg_crc32 and one["bytes"] == jpeg_bytes)
    elif "ijg" in cb:
        out["cpu_baseline"] = dict(cb["ijg"], kind="port", impl="IJG libjpeg 9d C API (non-SIMD; Pillow's libjpeg-turbo cannot write 4:4:0 / 4:1:1), all host cores")
        if "ijg_1core" in cb:
            out["cpu_baseline_1core"] = dict(cb["ijg_1core"], impl="IJG libjpeg 9d C API (non-SIMD), one core")
    else:
        out["cpu_baseline"] = dict(cb["port"], kind="port", impl="oracle/jpeg_oracle.c")
    out["cpu_oracle_port"] = dict(cb["port"], impl="oracle/jpeg_oracle.c (the checker, OpenMP-free, one strip per core)")
    return out


def golden_fingerprint(args, optimize, restart_interval):
    """(crc32 hex, bytes) of this configuration's file from tests/golden/big_8320x40000_q95.json (written by
    tests/make_golden_big.py on the CPU), or None when the run is not one of its configurations."""
    if (args.width, args.height, args.quality) != (W_IMG, H_IMG, QUALITY) or args.fmt not in ("bgr", "rgb"):
        return None
    path = os.path.join(ROOT, "tests", "golden", "big_8320x40000_q95.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    css = {"444": 0, "422": 1, "420": 2, "440": 3, "411": 4, "410": 5}[args.css]
    e = d.get("cases", {}).get("css%d_ri%d_%s" % (css, restart_interval, "progressive" if args.progressive else "opt" if optimize else "fix"))
    if not isinstance(e, dict) or "crc32" not in e:
        return None
    return (e["crc32"], e.get("len"))


def bpp_stage_b(css):
    """Algorithmic bytes per pixel of a separate histogram pass (SURVEY 8d stage B): read the int16 coefficients once."""
    f = {"444": 2.0, "422": 1.0, "440": 1.0, "420": 0.5, "411": 0.5, "410": 0.25}[css]
    return 2.0 * (1.0 + f)


def hbm_copy_ceiling(torch, mij, dev):
    """On-box streaming ceiling: device-to-device copy of 1 GiB with the library's own 16-B/lane copy kernel
    (MI355X_MICROARCH.md quotes ~6.3 TB/s for a float4 copy), read + write bytes per second, timed with events on the stream
    the kernel runs on. Second value: torch's copy_ of the same buffers, for reference (round 1 quoted that one)."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    a.zero_()
    st = torch.cuda.current_stream()

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        e1.synchronize()
        return 2 * n / (e0.elapsed_time(e1) * 1e-3 / reps) / 1e9

    own = timed(lambda: mij.copy_bench_device(b.data_ptr(), a.data_ptr(), n, st.cuda_stream))
    lib = timed(lambda: b.copy_(a))
    del a, b
    torch.cuda.empty_cache()
    return own, lib


def measured_traffic(kernel, args, optimize, world, lib_hash):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/hbm_traffic.py), or None if the passes were
    taken on a different workload than this run OR on a different build of the library than the one loaded now (the file
    records the source hash compiled into the library that ran under the counters). A process cannot read its own PMC
    counters; they come from rocprofv3."""
    import glob
    default = (args.width == W_IMG and args.height == H_IMG and args.css == CSS_NAME and args.quality == QUALITY and optimize and world == 1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not default or not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    rel = os.path.relpath(files[-1], ROOT)
    if d.get("library_source_hash") != lib_hash:
        return None, "%s was taken on another build of the library (%s...): not quoted" % (rel, str(d.get("library_source_hash"))[:12])
    k = d.get("kernels", {}).get(kernel)
    if k and "valu_wave_instructions" in k:
        # (the headline path's kernels only: k_transform_nostats is the stage-A-alone pass, not part of an image's step)
        _VALU[kernel] = {kk: vv["valu_wave_instructions"] for kk, vv in d["kernels"].items() if "valu_wave_instructions" in vv and kk != "k_transform_nostats"}
    return (k["total_bytes"], rel) if k else (None, None)


_VALU = {}


def valu_classes(lib_hash):
    """Per-kernel class mix of the vector instructions from profiles/r*_valu_classes.json (tools/valu_bound.py), only if it was made from
    the sources of the library loaded now."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_classes.json")))
    if not files:
        return {}
    try:
        with open(files[-1]) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return {}
    if d.get("library_source_hash") != lib_hash:
        return {}
    out = dict(d.get("kernels", {}))
    out["_source"] = os.path.relpath(files[-1], ROOT)
    return out


def _psnr_check(jpeg, W, H, fmt, d_img):
    """Decode with a stock decoder (libjpeg-turbo via Pillow) and compare with the source image, in bands."""
    import numpy as np
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    t0 = time.perf_counter()
    dec = np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))
    dec_crc = "%08x" % zlib.crc32(memoryview(np.ascontiguousarray(dec)).cast("B"))
    se, band = 0.0, 2000
    for y in range(0, H, band):
        n = min(band, H - y)
        if d_img is not None:
            src = d_img[y:y + n].cpu().numpy()
            if fmt == "bgr":
                src = src[..., ::-1]
        else:   # N > 1: rank 0 holds only its strip; regenerate the band with the library's device generator (not the oracle)
            import torch
            import nvjpeg_imagecompressor_amd as mij
            t = torch.empty((n, W, 3), dtype=torch.uint8, device="cuda")
            mij.synth_image_device(t.data_ptr(), W, y, n, W * 3, bgr=False, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            src = t.cpu().numpy()
        diff = dec[y:y + n].astype(np.int32) - src.astype(np.int32)
        se += float((diff * diff).sum())
    mse = se / (3.0 * W * H)
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    return round(psnr, 3), "stock decoder: Pillow/libjpeg-turbo, full image, %.1fs" % (time.perf_counter() - t0), dec_crc


if __name__ == "__main__":
    sys.exit(main())
