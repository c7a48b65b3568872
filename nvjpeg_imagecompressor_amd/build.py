"""Builds libmijpeg.so (HIP kernels + C ABI) in-tree with hipcc for gfx950. No GPU needed to build.

The library is linked WITHOUT a DT_NEEDED entry for libamdhip64: a process must contain exactly one HIP runtime, and
PyTorch wheels bundle their own copy next to the system one in /opt/rocm. Whoever loads libmijpeg.so provides the
runtime: `_lib.load()` binds it to torch's runtime when torch is already imported and to /opt/rocm's otherwise; a C++
program simply links `-lmijpeg -lamdhip64`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmijpeg.so")
SOURCES = ["mij_kernels.hip", "mij_api.hip", "mij_decode_api.hip"]
HEADERS = ["mij_internal.h", "k_common.inc", "k_transform.inc", "k_batch.inc", "k_stats.inc", "k_tables.inc", "k_encode.inc", "k_encode_prog.inc", "k_finish.inc", "k_synth.inc", "k_decode.inc", "k_decode_scans.inc", "k_decode_par.inc", "k_launch.inc", os.path.join("..", "..", "include", "mi_jpeg.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + HEADERS
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in deps) or os.path.getmtime(__file__) > t


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    link = ["g++", "-shared", "-o", LIB] + objs  # no -lamdhip64 on purpose (see module docstring)
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
