#!/usr/bin/env python3
"""CPU baselines for bench.py, run in their own interpreter (no torch / HIP in this process, so a
fork-based pool is safe). Legs, each on a BOUNDED sample of the bench's workload:

  turbo        libjpeg-turbo 3.1.4.1 (through Pillow), ALL cores: one strip of the synthetic image per worker process
  turbo_1core  the same library on ONE core, one image of `--one-core-rows` rows (default: the whole image: the library's
               native mode)
  port         the oracle C port (oracle/jpeg_oracle.c), all cores
  ijg          IJG libjpeg 9d through its C API (oracle/ijg_harness, non-SIMD), all cores + 1 core: the only stock encoder
               in this image for 4:4:0 / 4:1:1 (Pillow maps those samplings to 4:2:0)

Prints one JSON object keyed by leg."""
import argparse
import io
import json
import os
import subprocess
import sys
import tempfile
import time
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_S = {}
HS_VS = {0: (1, 1), 1: (2, 1), 2: (2, 2), 3: (1, 2), 4: (4, 1), 5: (4, 2)}


def _init(width, height, rows, idx_base):
    from oracle import oracle as O
    _S["O"] = O
    _S["args"] = (width, height, rows)


def _strip(i):
    width, height, rows = _S["args"]
    if ("img", i) not in _S:
        _S[("img", i)] = _S["O"].synth_rgb(width, height, y0=(i * rows) % max(1, height - rows), rows=rows)
    return _S[("img", i)]


def _turbo(img, quality, css, optimize, ri, keep=None):
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    b = io.BytesIO()
    Image.frombuffer("RGB", (img.shape[1], img.shape[0]), img, "raw", "RGB", 0, 1).save(
        b, "JPEG", quality=quality, subsampling=css, optimize=optimize, restart_marker_blocks=ri)
    if keep is not None:
        keep.append(b)          # the caller fingerprints the file outside its timed region
    return b.tell()


def _run(job):
    kind, i, quality, css, optimize, ri = job
    img = _strip(i)
    t0 = time.perf_counter()
    if kind == "port":
        n = len(_S["O"].encode(img, quality, css, optimize, ri))
    elif kind == "turbo":
        n = _turbo(img, quality, css, optimize, ri)
    else:   # ijg: the harness reads the strip from a file in /dev/shm (outside its timed region) and reports its own time
        with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, suffix=".raw") as f:
            f.write(img.tobytes())
            f.flush()
            hs, vs = HS_VS[css]
            out = subprocess.check_output([os.path.join(ROOT, "oracle", "ijg_harness"), "bench", f.name, str(img.shape[1]), str(img.shape[0]),
                                           str(quality), str(hs), str(vs), str(int(optimize)), str(ri), "1"], text=True).split()
        return int(out[1]), float(out[0])
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--rows", type=int, default=1000)
    ap.add_argument("--one-core-rows", type=int, default=0, help="rows of the 1-core sample (0 = the whole image)")
    ap.add_argument("--cores", type=int, required=True)
    ap.add_argument("--quality", type=int, default=95)
    ap.add_argument("--css", type=int, default=1)
    ap.add_argument("--optimize", type=int, default=1)
    ap.add_argument("--ri", type=int, default=104)
    a = ap.parse_args()
    out = {}
    what = "q%d css%d %s DRI=%d" % (a.quality, a.css, "optimised" if a.optimize else "fixed", a.ri)
    kinds = ["port"] + (["turbo"] if a.css <= 2 else []) + (["ijg"] if a.css in (3, 4) and os.path.exists(os.path.join(ROOT, "oracle", "ijg_harness")) else [])
    with Pool(a.cores, initializer=_init, initargs=(a.width, a.height, a.rows, 0)) as pool:
        pool.map(_strip, range(a.cores), chunksize=1)      # generate the strips (untimed)
        for kind in kinds:
            best, sizes = None, None
            for _ in range(2):
                t0 = time.perf_counter()
                res = pool.map(_run, [(kind, i, a.quality, a.css, bool(a.optimize), a.ri) for i in range(a.cores)], chunksize=1)
                dt = time.perf_counter() - t0
                if kind == "ijg":
                    dt = max(r[1] for r in res)     # the harness's own clocks (process start-up and file reading excluded)
                if best is None or dt < best:
                    best, sizes = dt, [r[0] for r in res]
            mpix = a.cores * a.rows * a.width / 1e6
            out[kind] = {"value": round(mpix / best, 2), "unit": "Mpixels/s", "cores": a.cores, "bytes": int(sum(sizes)),
                         "sample": "%d strips of %dx%d synthetic RGB8, one per core in %d worker processes, %s, best of 2"
                                   % (a.cores, a.width, a.rows, a.cores, what)}
    # ---- one core: the library's native mode ----
    rows1 = a.one_core_rows or a.height
    _init(a.width, a.height, rows1, 0)
    img = _S["O"].synth_rgb(a.width, a.height, y0=0, rows=rows1)
    if a.css <= 2:
        _turbo(img[:64], a.quality, a.css, bool(a.optimize), a.ri)    # warm-up (library load, table init)
        keep = []
        t0 = time.perf_counter()
        n = _turbo(img, a.quality, a.css, bool(a.optimize), a.ri, keep)
        dt = time.perf_counter() - t0
        import zlib
        out["turbo_1core"] = {"value": round(a.width * rows1 / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": 1, "bytes": n,
                              "crc32": "%08x" % zlib.crc32(keep[0].getbuffer()), "whole_image": rows1 == a.height,
                              "sample": "one %dx%d synthetic RGB8 image on one core, %s, one run after a warm-up" % (a.width, rows1, what)}
    elif "ijg" in out:
        rows1 = min(rows1, 4 * a.rows)      # non-SIMD: keep the sample bounded
        _S[("img", 0)] = img[:rows1]
        n, dt = _run(("ijg", 0, a.quality, a.css, bool(a.optimize), a.ri))
        out["ijg_1core"] = {"value": round(a.width * rows1 / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": 1, "bytes": n,
                            "sample": "one %dx%d synthetic RGB8 image on one core, %s" % (a.width, rows1, what)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
