#!/usr/bin/env python3
"""CPU baselines for bench.py, run in their own interpreter (no torch / HIP in this process, so a fork-based pool is
safe): the oracle C port and libjpeg-turbo (through Pillow), one strip of the synthetic image per worker process.
Prints one JSON object:  {"port": {...}, "turbo": {...}}"""
import argparse
import io
import json
import os
import sys
import time
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_S = {}


def _init(width, height, rows, idx_base):
    from oracle import oracle as O
    _S["O"] = O
    _S["args"] = (width, height, rows)


def _strip(i):
    width, height, rows = _S["args"]
    if ("img", i) not in _S:
        _S[("img", i)] = _S["O"].synth_rgb(width, height, y0=(i * rows) % max(1, height - rows), rows=rows)
    return _S[("img", i)]


def _run(job):
    kind, i, quality, css, optimize, ri = job
    img = _strip(i)
    t0 = time.perf_counter()
    if kind == "port":
        n = len(_S["O"].encode(img, quality, css, optimize, ri))
    else:
        from PIL import Image
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=quality, subsampling=css, optimize=optimize, restart_marker_blocks=ri)
        n = len(b.getvalue())
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--rows", type=int, default=1000)
    ap.add_argument("--cores", type=int, required=True)
    ap.add_argument("--quality", type=int, default=95)
    ap.add_argument("--css", type=int, default=1)
    ap.add_argument("--optimize", type=int, default=1)
    ap.add_argument("--ri", type=int, default=104)
    a = ap.parse_args()
    out = {}
    with Pool(a.cores, initializer=_init, initargs=(a.width, a.height, a.rows, 0)) as pool:
        pool.map(_strip, range(a.cores), chunksize=1)      # generate the strips (untimed)
        for kind in ("port", "turbo"):
            if kind == "turbo" and a.css > 2:
                continue    # Pillow's libjpeg-turbo build cannot produce 4:4:0 / 4:1:1
            best, sizes = None, None
            for _ in range(2):
                t0 = time.perf_counter()
                res = pool.map(_run, [(kind, i, a.quality, a.css, bool(a.optimize), a.ri) for i in range(a.cores)], chunksize=1)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, sizes = dt, [r[0] for r in res]
            mpix = a.cores * a.rows * a.width / 1e6
            out[kind] = {"value": round(mpix / best, 2), "unit": "Mpixels/s", "cores": a.cores, "bytes": int(sum(sizes)),
                         "sample": "%d strips of %dx%d synthetic RGB8, one per core in %d worker processes, q%d css%d %s DRI=%d, "
                                   "best of 2" % (a.cores, a.width, a.rows, a.cores, a.quality, a.css,
                                                  "optimised" if a.optimize else "fixed", a.ri)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
