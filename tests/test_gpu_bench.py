"""bench.py on one GPU (round 4): whatever loop stands behind `value`, the file is the same file, the per-kernel figures come from
the separate one-stream pass, and the line carries the stage-A-alone roofline, the measured clock and the single-image time."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MIJ_BENCH_CHILD", "MIJ_BENCH_DIRECT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


SMALL = ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-psnr", "--height", "4000", "--kernel-pass", "2", "--stage-a-pass", "2"]


@pytest.fixture(scope="module")
def default_line():
    return _bench(SMALL)


def test_default_loop_overlaps_and_reports_the_separate_pass(default_line):
    d = default_line
    assert d["loop"] == "overlap" and d["config"]["streams"] == 2 and d["config"]["images_in_flight"] == 3
    assert d["per_kernel_pass"] == dict(d["per_kernel_pass"], images=2, streams=1, file_identical_to_timed_loop=True)
    assert "one-stream pass" in d["roofline"]["note"] and d["roofline"]["kernel"] == "k_transform"
    assert abs(d["roofline"]["avg_launch_ms"] - d["stage_ms"]["transform"]) < 1e-3
    a = d["roofline"]["stage_A_alone"]
    assert a["launches"] == 2 and a["algorithmic_bytes_per_launch"] == 7 * 8320 * 4000 and 0 < a["frac"] < 1
    assert a["avg_launch_ms"] < d["roofline"]["avg_launch_ms"]
    when = [p["when"] for p in d["clock"]["probes"]]
    assert when[0] == "start (device idle before)" and when[1] == "after the warm-up steps" and when[-3:] == [
        "right after the timed region", "after the per-kernel pass", "after the stage-A pass"]
    assert sum("right before the timed region" in w for w in when) == 1 and d["clock"]["settle_steps"] == 10 * sum("settle round" in w for w in when)
    assert all(500 < p["valu_MHz"] < 3000 and 500 < p["counter_MHz"] < 3000 for p in d["clock"]["probes"])
    assert "valu_issue" not in d["roofline"] or "clock_MHz" in d["roofline"]["valu_issue"]
    assert d["single_image_ms"] > 0 and d["stage_ms"]["total"] > 0


@pytest.mark.parametrize("loop", ["tables-ahead", "one-stream", "two-streams"])
def test_every_loop_writes_the_same_file(default_line, loop):
    d = _bench(SMALL + ["--loop", loop])
    assert d["loop"] == loop and (d["jpeg_crc32"], d["jpeg_bytes"]) == (default_line["jpeg_crc32"], default_line["jpeg_bytes"])
    assert d["per_kernel_pass"]["file_identical_to_timed_loop"] is True


def test_old_spellings_still_select_their_loops():
    assert _bench(SMALL + ["--two-streams"])["loop"] == "two-streams"
    assert _bench(SMALL + ["--no-tables-ahead"])["loop"] == "one-stream"


def test_fixed_tables_and_progressive_lines():
    d = _bench(SMALL + ["--no-optimize"])
    assert d["loop"] == "overlap" and d["per_kernel_pass"]["file_identical_to_timed_loop"] is True
    assert d["roofline"]["stage_A_alone"]["avg_launch_ms"] > 0          # the headline kernel IS stage A alone here
    p = _bench(SMALL + ["--progressive"])
    assert p["loop"] == "progressive" and "stage_A_alone" not in p["roofline"] and p["per_kernel_pass"]["file_identical_to_timed_loop"] is True


def test_driver_line_carries_every_single_gpu_config_and_both_decoders():
    """Round 5: the default workload's line holds BASELINE configs 3 and 5 and both decoders, each checked against a committed golden or
    against a stock decoder inside the run (the CPU legs are switched off here; the driver's command runs them too)."""
    d = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--kernel-pass", "2", "--stage-a-pass", "2"], timeout=900)
    assert d["golden_match"] is True and d["untimed_steps"] >= 1
    rows = d["configs"]["table1"]["rows"]
    assert sorted(rows) == ["411", "420", "422", "440", "444"]
    for css, r in rows.items():
        assert r["golden_match"] is True and r["steps"] >= 10 and 0.3 < r["K1_frac"] < 1 and 0.2 < r["K4_frac"] < 1 and r["decode_ms"] > 0, (css, r)
    assert abs(rows["422"]["psnr_db"] - 31.162) < 0.01 and rows["422"]["crc32"] == d["jpeg_crc32"]
    sec = d["configs"]["secondary"]
    assert sec["first_layer"]["golden_match"] is True
    for key in ("q95_css1_gain1", "q98_css0_gain1"):
        c = sec["cases"][key]
        assert c["golden_match"] is True and c["secondary_compress_ms"] > 0 and c["decode_both_and_add_ms"] > 0
        assert [round(x, 2) for x in (c["psnr_first_layer_db"], c["psnr_both_layers_db"])] == [round(x, 2) for x in c["golden_psnr_db"]]
    own = d["decode"]["own_file"]
    assert own["identical_to_pillow"] is True and own["roofline"]["algorithmic_bytes"] == d["jpeg_bytes"] + 3 * 8320 * 40000 and 0 < own["roofline"]["frac"] < 1
    prog = d["decode"]["progressive_nodri"]
    assert prog["golden_match"] is True and prog["scans_parallel"] == prog["scans_tried"] == 9 and prog["device_ms"] < 1000
    v = d["roofline"].get("valu_issue")
    if v is not None and v["model"] == "bound":          # (only when profiles/ holds the class file of THIS build)
        k = v["kernels"]["k_transform"]
        assert k["bound_ms"] < k["launch_ms"] and 2.0 <= k["bound_cycles_per_instruction"] <= 4.0 <= k["cycles_per_valu_instruction"] + 1.0
