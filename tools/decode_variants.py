#!/usr/bin/env python3
"""Builds libmijpeg.so with other compile-time settings of the parallel Huffman decoder (k_decode_par.inc) and, with `run`,
times the full-size decode with each (tools/decode_fullsize.py).  Usage:
    python tools/decode_variants.py build          (here, no GPU needed)
    python tools/decode_variants.py run            (on the GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nvjpeg_imagecompressor_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants")
VARIANTS = {"dec_default": {}, "dec_s256": {"MIJ_PAR_S": 256}, "dec_s384": {"MIJ_PAR_S": 384}, "dec_s512": {"MIJ_PAR_S": 512}, "dec_s640": {"MIJ_PAR_S": 640},
            "dec_s768": {"MIJ_PAR_S": 768}, "dec_s2048": {"MIJ_PAR_S": 2048},
            # write pass with 128 / 256 / 512 lanes per workgroup (the Huffman tables in LDS are per workgroup): tools/decode_hammer.py
            "dec_wg128": {"MIJ_PAR_WG2": 128}, "dec_wg256": {"MIJ_PAR_WG2": 256}, "dec_wg512": {"MIJ_PAR_WG2": 512},
            # round 3, the symbol loop: limit compare for long codes; wider look-ahead tables shared by larger workgroups
            "dec_lim": {"MIJ_PAR_LIMITS": 1},
            # wider look-ahead for the speculative / synchronisation passes only (the write pass keeps 9 bits and 64 lanes)
            "dec01_lb10_128": {"MIJ_PAR_LOOK_BITS": 10, "MIJ_PAR_WG01": 128}, "dec01_lb11_256": {"MIJ_PAR_LOOK_BITS": 11, "MIJ_PAR_WG01": 256},
            "dec01_lb12_512": {"MIJ_PAR_LOOK_BITS": 12, "MIJ_PAR_WG01": 512}, "dec01_lb11_128": {"MIJ_PAR_LOOK_BITS": 11, "MIJ_PAR_WG01": 128},
            "dec01_lb10_64": {"MIJ_PAR_LOOK_BITS": 10},
            # the same for the write pass (on top of the 11-bit / 256-lane default of the other two)
            "dec2_lb10_64": {"MIJ_PAR_LOOK_BITS2": 10}, "dec2_lb10_128": {"MIJ_PAR_LOOK_BITS2": 10, "MIJ_PAR_WG2": 128},
            "dec2_lb11_256": {"MIJ_PAR_LOOK_BITS2": 11, "MIJ_PAR_WG2": 256}, "dec2_lb9_128": {"MIJ_PAR_WG2": 128},
            # EXPERIMENTS with wrong output: what the write pass's global stores cost
            "dec_nostore": {"MIJ_PAR_NOSTORE": 1}, "dec_nozero": {"MIJ_PAR_NOSTORE": 2},
            "dec_stg32": {"MIJ_PAR_STG": 32},
            "dec_all256": {"MIJ_PAR_WG01": 256, "MIJ_PAR_WG2": 256},
            "dec_lb10_128": {"MIJ_PAR_LOOK_BITS": 10, "MIJ_PAR_LIMITS": 1, "MIJ_PAR_WG01": 128, "MIJ_PAR_WG2": 128},
            "dec_lb10_256": {"MIJ_PAR_LOOK_BITS": 10, "MIJ_PAR_LIMITS": 1, "MIJ_PAR_WG01": 256, "MIJ_PAR_WG2": 256},
            "dec_lb11_256": {"MIJ_PAR_LOOK_BITS": 11, "MIJ_PAR_LIMITS": 1, "MIJ_PAR_WG01": 256, "MIJ_PAR_WG2": 256},
            "dec_lb11_512": {"MIJ_PAR_LOOK_BITS": 11, "MIJ_PAR_LIMITS": 1, "MIJ_PAR_WG01": 512, "MIJ_PAR_WG2": 512},
            "dec_lb12_512": {"MIJ_PAR_LOOK_BITS": 12, "MIJ_PAR_LIMITS": 1, "MIJ_PAR_WG01": 512, "MIJ_PAR_WG2": 512}}
if os.environ.get("MIJ_VARIANTS"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["MIJ_VARIANTS"].split(",")}


def build():
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda kv: build_one(*kv), VARIANTS.items()))


def build_one(name, defs):
    if True:
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        objs = []
        for src in ("mij_kernels.hip", "mij_api.hip", "mij_decode_api.hip"):
            obj = os.path.join(d, src.replace(".hip", ".o"))
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DMIJ_FAST_BUILD"] + [
                "-D%s=%s" % kv for kv in defs.items()] + ["-c", os.path.join(CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                print(r.stderr[-2000:])
                raise SystemExit(1)
            objs.append(obj)
        subprocess.check_call(["g++", "-shared", "-o", os.path.join(d, "libmijpeg.so")] + objs)
        print("built", name, flush=True)


def run():
    for name in VARIANTS:
        env = dict(os.environ, MIJ_LIB_PATH=os.path.join(OUT, name, "libmijpeg.so"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "decode_fullsize.py")], capture_output=True, text=True, env=env)
        trace = [l.split(": ")[1].split()[0] for l in r.stderr.splitlines() if l.startswith("[par]")]
        print(name, r.stdout.strip()[-330:] if r.returncode == 0 else "FAILED " + r.stderr[-600:], "redo counts per pass:", trace[-4:], flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
