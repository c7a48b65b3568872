"""The supervisor of `python bench.py --gpus N` (bench.py: supervise / _run_attempt) without a GPU: child ranks are played by
tiny scripts that speak the children's protocol on stdout ("##progress ..." marks, one JSON line). What is under test is the
launcher's behaviour on first contact with a node: relay of rank 0's line, exit status, the fallback ladder
put -> sendrecv -> serial with FRESH processes after a rank died or the run stalled, and that only its own children are killed."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys, time
rank, world, mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
plan = json.loads(os.environ["FAKE_PLAN"]).get(mode, "ok")
open(os.environ["FAKE_LOG"], "a").write("%s %d %d\n" % (mode, rank, os.getpid()))
assert os.environ["MIJ_BENCH_CHILD"] == "1" and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
if rank == 0:
    print("##progress imported", flush=True)
if plan == "die":
    if rank == 1:
        os._exit(3)
    time.sleep(600)          # the others wait for a collective that never completes
if plan == "hang":
    if rank == 0:
        print("##progress process group up", flush=True)
    time.sleep(600)
if plan == "hang_teardown":
    if rank == 0:
        print(json.dumps({"value": 1.0, "config": {"gather": mode}}), flush=True)
    time.sleep(600)
if rank == 0:
    print("stray line", flush=True)
    print(json.dumps({"value": 1.0, "n_gpus": world, "config": {"gather": mode}}), flush=True)
'''


@pytest.fixture
def bench(monkeypatch, tmp_path):
    import importlib
    b = importlib.import_module("bench")
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    monkeypatch.setattr(b, "_child_argv", lambda mode: [sys.executable, str(script), mode])
    monkeypatch.setenv("FAKE_LOG", str(tmp_path / "log.txt"))
    monkeypatch.setenv("MIJ_BENCH_WATCHDOG_S", "3")
    monkeypatch.setenv("MIJ_BENCH_WATCHDOG_INIT_S", "30")
    b._log = str(tmp_path / "log.txt")
    return b


class Args:
    gather, no_fallback, progressive = "put", False, False


def _run(bench, capsys, plan, monkeypatch, world=3, **kw):
    monkeypatch.setenv("FAKE_PLAN", json.dumps(plan))
    a = Args()
    for k, v in kw.items():
        setattr(a, k, v)
    rc = bench.supervise(a, world, "test")
    out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    log = [l.split() for l in open(bench._log).read().splitlines()]
    return rc, (json.loads(out[-1]) if out else None), log


def _gone(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    # a zombie of ours still answers kill(0): reap-state check
    try:
        return open("/proc/%d/stat" % pid).read().split()[2] == "Z"
    except OSError:
        return True


def test_clean_run_relays_rank0_line(bench, capsys, monkeypatch):
    rc, d, log = _run(bench, capsys, {}, monkeypatch)
    assert rc == 0 and d["config"]["gather"] == "put" and d["gather_fallback"] is None and d["launcher"] == "test"
    assert sorted(int(r) for m, r, _ in log) == [0, 1, 2] and {m for m, _, _ in log} == {"put"}


def test_auto_runs_both_gathers_and_prints_the_faster(bench, capsys, monkeypatch):
    """--gather auto (the default): after the put pipeline's run, a complete run with send/recv; both values in the line."""
    rc, d, log = _run(bench, capsys, {}, monkeypatch, gather="auto")
    assert rc == 0 and set(d["gather_runs"]) == {"put", "sendrecv"} and d["config"]["gather"] in ("put", "sendrecv")
    assert {m for m, _, _ in log} == {"put", "sendrecv"} and len(log) == 6
    rc, d, log = _run(bench, capsys, {"sendrecv": "die"}, monkeypatch, gather="auto")          # the alternative failing costs nothing
    assert rc == 0 and d["config"]["gather"] == "put" and "failed" in d["gather_runs"]["sendrecv"]


def test_dead_rank_falls_back_with_fresh_processes(bench, capsys, monkeypatch):
    rc, d, log = _run(bench, capsys, {"put": "die"}, monkeypatch)
    assert rc == 0 and d["config"]["gather"] == "sendrecv"
    assert "put: rank 1 exited with status 3" in d["gather_fallback"]
    pids = {}
    for m, r, pid in log:
        pids.setdefault(m, set()).add(int(pid))
    assert len(pids["put"]) == 3 and len(pids["sendrecv"]) == 3 and not (pids["put"] & pids["sendrecv"])     # fresh processes
    assert all(_gone(p) for p in pids["put"])


def test_stalled_run_is_killed_and_the_ladder_descends_to_serial(bench, capsys, monkeypatch):
    t0 = time.monotonic()
    rc, d, log = _run(bench, capsys, {"put": "hang", "sendrecv": "die"}, monkeypatch)
    assert rc == 0 and d["config"]["gather"] == "serial"
    assert "put: no progress for 3 s after 'process group up'" in d["gather_fallback"] and "sendrecv: rank 1 exited" in d["gather_fallback"]
    assert time.monotonic() - t0 < 60
    assert all(_gone(int(pid)) for m, _, pid in log if m == "put")         # the hung ranks were ended, exactly those


def test_every_gather_failing_is_an_error_not_a_line(bench, capsys, monkeypatch):
    rc, d, _ = _run(bench, capsys, {"put": "die", "sendrecv": "die", "serial": "die"}, monkeypatch)
    assert rc == 1 and d is None
    rc, d, _ = _run(bench, capsys, {"put": "die"}, monkeypatch, no_fallback=True)
    assert rc == 1 and d is None


def test_result_stands_when_a_rank_hangs_in_teardown(bench, capsys, monkeypatch):
    rc, d, log = _run(bench, capsys, {"put": "hang_teardown"}, monkeypatch, world=2)
    assert rc == 0 and d["config"]["gather"] == "put" and "ended by the supervisor" in d["teardown_note"]
    assert all(_gone(int(pid)) for _, _, pid in log)


def test_a_terminated_supervisor_takes_its_children_with_it(tmp_path):
    """SIGTERM to `python bench.py --gpus N` (a driver's timeout, say) must not leave ranks behind: they lead their own
    sessions, so nothing but the supervisor would end them."""
    import signal
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    log = tmp_path / "log.txt"
    drv = tmp_path / "drv.py"
    drv.write_text("import sys, json\nsys.path.insert(0, %r)\nimport bench\nbench._child_argv = lambda mode: [sys.executable, %r, mode]\n"
                   "class A: gather, no_fallback, progressive = 'put', False, False\nsys.exit(bench.supervise(A(), 3, 'test'))\n" % (ROOT, str(script)))
    p = subprocess.Popen([sys.executable, str(drv)], env=dict(os.environ, FAKE_PLAN=json.dumps({"put": "hang"}), FAKE_LOG=str(log)),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t_end = time.monotonic() + 30
    while time.monotonic() < t_end and (not log.exists() or len(log.read_text().splitlines()) < 3):
        time.sleep(0.1)
    pids = [int(l.split()[2]) for l in log.read_text().splitlines()]
    assert len(pids) == 3
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    t_end = time.monotonic() + 10
    while time.monotonic() < t_end and not all(_gone(q) for q in pids):
        time.sleep(0.1)
    assert all(_gone(q) for q in pids)


def test_plain_command_line_needs_no_launcher():
    """`python bench.py --gpus 2` on a box without a GPU: the supervisor itself runs (no 'needs torch.distributed.run' exit),
    starts ranks, sees them fail for lack of a device and reports that -- without ever importing torch itself."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the N > 1 path is exercised by tests/test_gpu_sharded.py")
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-fallback"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 1 and "gather 'put' failed" in r.stderr
    # -X importtime is a flag of the PARENT interpreter only (the children are started without it): its import log is the parent's
    imported = [l.split("|")[-1].strip() for l in r.stderr.splitlines() if l.startswith("import time:")]
    assert imported and "torch" not in imported and "nvjpeg_imagecompressor_amd" not in imported


# ---- round 4: pieces of the one-GPU line that need no GPU ---------------------------------------------------------------
class _A:
    loop, two_streams, tables_ahead, progressive = None, False, None, False


def test_one_gpu_loop_selection():
    import importlib
    b = importlib.import_module("bench")
    a = _A()
    assert b._one_gpu_loop(a, True) == "overlap" and b._one_gpu_loop(a, False) == "overlap"          # the default, fixed tables too
    a.two_streams = True
    assert b._one_gpu_loop(a, True) == "two-streams"
    a.two_streams, a.tables_ahead = False, False
    assert b._one_gpu_loop(a, True) == "one-stream"
    a.tables_ahead = True
    assert b._one_gpu_loop(a, True) == "tables-ahead" and b._one_gpu_loop(a, False) == "one-stream"  # nothing to build ahead with fixed tables
    a.loop = "tables-ahead"
    a.progressive = True
    assert b._one_gpu_loop(a, True) == "progressive"


def test_cpu_legs_compare_the_whole_file_by_fingerprint(monkeypatch):
    """cpu_baseline_fields: libjpeg-turbo's whole-image file (one core, same run) against the GPU's by CRC + length; a partial
    one-core sample claims nothing."""
    import importlib
    b = importlib.import_module("bench")
    cb = {"turbo": {"value": 600.0, "unit": "Mpixels/s", "cores": 16, "bytes": 1, "sample": "s"},
          "turbo_1core": {"value": 60.0, "unit": "Mpixels/s", "cores": 1, "bytes": 203772997, "crc32": "47e0cdfa", "whole_image": True, "sample": "s"},
          "port": {"value": 300.0, "unit": "Mpixels/s", "cores": 16, "bytes": 1, "sample": "s"}}
    monkeypatch.setattr(b, "cpu_baselines", lambda args, optimize, ri: cb)
    out = b.cpu_baseline_fields(None, True, 64, "47e0cdfa", 203772997)
    assert out["turbo_file_identical"] is True and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline_1core"]["crc32"] == "47e0cdfa"
    assert b.cpu_baseline_fields(None, True, 64, "deadbeef", 203772997)["turbo_file_identical"] is False
    cb["turbo_1core"]["whole_image"] = False
    assert "turbo_file_identical" not in b.cpu_baseline_fields(None, True, 64, "47e0cdfa", 203772997)


def test_supervisor_adds_the_cpu_legs_at_n_gt_1(bench, capsys, monkeypatch):
    """N > 1: the children never run the CPU legs (they hold the GPUs); the supervisor does, after they are gone."""
    monkeypatch.setattr(bench, "cpu_baseline_fields", lambda args, optimize, ri, crc, nbytes: {"cpu_baseline": {"value": 1.0, "kind": "port", "crc_seen": crc}})
    child = CHILD.replace('print(json.dumps({"value": 1.0, "n_gpus": world, "config": {"gather": mode}}), flush=True)',
                          'print(json.dumps({"value": 1.0, "n_gpus": world, "jpeg_crc32": "47e0cdfa", "jpeg_bytes": 5, "config": {"gather": mode, "restart_interval": 64}}), flush=True)')
    import pathlib
    script = pathlib.Path(bench._log).parent / "child2.py"
    script.write_text(child)
    monkeypatch.setattr(bench, "_child_argv", lambda mode: [sys.executable, str(script), mode])
    rc, line, log = _run(bench, capsys, {}, monkeypatch, world=2, no_cpu_baseline=False, no_optimize=False, gather="sendrecv")
    assert rc == 0 and line["cpu_baseline"]["crc_seen"] == "47e0cdfa"
    rc, line, log = _run(bench, capsys, {}, monkeypatch, world=2, no_cpu_baseline=True, no_optimize=False, gather="sendrecv")
    assert rc == 0 and "cpu_baseline" not in line
