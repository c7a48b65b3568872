#!/usr/bin/env python3
"""Builds experiment variants of libmijpeg.so (compile-time switches in csrc/k_common.inc / k_transform.inc) and, with
`run`, times the transform stage of each on the GPU through bench.py.  Usage:
    python tools/k1_variants.py build            (here, no GPU needed)
    python tools/k1_variants.py run [--no-optimize]   (on the GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nvjpeg_imagecompressor_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants")
VARIANTS = {   # experiment switches of k_common.inc / k_transform.inc; "default" is what ships (K4's strip sizes are template
    # parameters now: MIJ_K4_WIDE=1 in the environment selects the 24-word kernel at run time)
    "default": {},
    # round 3: the instruction-count changes, one at a time (defaults: DOT3 on, RTZ and STAT2 off; profiles/r03_k1_variants.txt)
    "no_dot3": {"MIJ_K1_DOT3": 0},
    "rtz": {"MIJ_K1_RTZ": 1},
    "stat2_c5": {"MIJ_K1_STAT2": 1},
    "stat2_c4": {"MIJ_K1_STAT2": 1, "MIJ_HIST_COPIES": 4},
    "all3_c4": {"MIJ_K1_STAT2": 1, "MIJ_HIST_COPIES": 4, "MIJ_K1_RTZ": 1},
    "nt_loads": {"MIJ_K1_NT_LOADS": 1},
    "k4_nt_loads": {"MIJ_COEF_NT_LOADS": 1},
    "rtz_dot3": {"MIJ_K1_RTZ": 1, "MIJ_K1_DOT3": 1},
    "sched_ilp": {"_FLAGS": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]},
    "sched_mem": {"_FLAGS": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]},
    # (-amdgpu-sched-strategy=iterative-ilp crashes this hipcc; max-ilp / max-memory-clause: K1 unchanged, K4 4 % slower with scratch)
    "waves2": {"MIJ_K1_WAVES": 2},
    "copies2": {"MIJ_HIST_COPIES": 2},
    "no_atomics": {"MIJ_K1_STATMODE": 1},
    "conflict_free": {"MIJ_K1_STATMODE": 2},
    "nostore": {"MIJ_K1_NOSTORE": 1},
    "noload": {"MIJ_K1_NOLOAD": 1},
    "waves4": {"MIJ_K1_WAVES": 4},
    "nz_from_size": {"MIJ_K1_NZ_FROM_SIZE": 1},
    "noflush": {"MIJ_K1_NOFLUSH": 1},
    "nt_stores": {"MIJ_K1_NT_STORES": 1},
    # round 4: rows 0..3 of the next pass requested before phase 2 (default on)
    "no_prefetch": {"MIJ_K1_PREFETCH": 0},
    "lds444": {"MIJ_K1_444_REGS": 0},            # 4:4:4 chroma through LDS (rounds 1-3)
    "lds444_no_prefetch": {"MIJ_K1_444_REGS": 0, "MIJ_K1_PREFETCH": 0},
    "no_pk_fma": {"MIJ_K1_PK_FMA": 0},            # colour conversion with scalar FMAs (rounds 1-4a)
    "px_g1024": {"MIJ_PX_G": 1024},
    "px_g512": {"MIJ_PX_G": 512},
    "copies444_7": {"MIJ_HIST_COPIES_444": 7},
    "copies444_8": {"MIJ_HIST_COPIES_444": 8},
    "copies444_6": {"MIJ_HIST_COPIES_444": 6},
}
if os.environ.get("MIJ_VARIANTS"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["MIJ_VARIANTS"].split(",")}


def build():
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(int(os.environ.get("MIJ_VARIANTS_JOBS", "4"))) as ex:
        list(ex.map(lambda kv: build_one(*kv), VARIANTS.items()))


def build_one(name, defs):
    if True:
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        objs = []
        for src in ("mij_kernels.hip", "mij_api.hip", "mij_decode_api.hip"):
            obj = os.path.join(d, src.replace(".hip", ".o"))
            flags = defs.get("_FLAGS", [])           # a variant may also carry extra compiler flags
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + ([] if os.environ.get("MIJ_VARIANTS_FULL") else ["-DMIJ_FAST_BUILD"]) + [
                   "-Rpass-analysis=kernel-resource-usage"] + flags + ["-D%s=%s" % kv for kv in defs.items() if kv[0] != "_FLAGS"] + ["-c", os.path.join(CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                print(r.stderr[-2000:])
                raise SystemExit(1)
            if src == "mij_kernels.hip":
                lines = r.stderr.splitlines()
                for i, l in enumerate(lines):
                    if "Function Name: _ZN3mij8k_encodeILi1ELb0" in l:
                        info = [x.split("remark:")[1].split("[-R")[0].strip() for x in lines[i + 1:i + 12] if "VGPRs:" in x or "ScratchSize" in x or "Occupancy" in x or "LDS" in x]
                        print(name, "k_encode", info)
                    if "Function Name: _ZN3mij11k_transformILi2ELi1ELb1ELb" in l:
                        info = [x.split("remark:")[1].split("[-R")[0].strip() for x in lines[i + 1:i + 12] if "VGPRs:" in x or "ScratchSize" in x or "Occupancy" in x]
                        print(name, "stats" if "Lb1ELb1" in l else "nostats", info)
            objs.append(obj)
        subprocess.check_call(["g++", "-shared", "-o", os.path.join(d, "libmijpeg.so")] + objs)


def run(extra):
    # `prev`: a library built by hand from an earlier commit, for A/B runs on the same box
    for name in list(VARIANTS) + [d for d in ("prev",) if os.path.exists(os.path.join(OUT, d, "libmijpeg.so"))]:
        lib = os.path.join(OUT, name, "libmijpeg.so")
        env = dict(os.environ, MIJ_LIB_PATH=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-psnr"] + extra,
                           capture_output=True, text=True, env=env)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-20s transform %.4f ms  entropy %.4f  total %.4f  crc %s" % (name, d["stage_ms"]["transform"], d["stage_ms"]["entropy"], d["ms_per_step"], d["jpeg_crc32"]), flush=True)
        except Exception:
            print(name, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:])
