"""Builds libmijpeg.so (HIP kernels + C ABI) in-tree with hipcc for gfx950. No GPU needed to build.

The library is linked WITHOUT a DT_NEEDED entry for libamdhip64: a process must contain exactly one HIP runtime, and
PyTorch wheels bundle their own copy next to the system one in /opt/rocm. Whoever loads libmijpeg.so provides the
runtime: `_lib.load()` binds it to torch's runtime when torch is already imported and to /opt/rocm's otherwise; a C++
program simply links `-lmijpeg -lamdhip64`.

Staleness is decided by CONTENT, not by mtime: `source_hash()` is the SHA-256 over every source, header and the compile
flags; it is compiled into the library (`mij_source_hash()`), so a shipped .so can be checked against the tree it is
supposed to come from (tests/test_abi.py does), and `needs_build()` compares it with the hash recorded next to the
library. Objects are cached per translation unit the same way.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmijpeg.so")
STAMP = LIB + ".srchash"
SOURCES = ["mij_kernels.hip", "mij_api.hip", "mij_decode_api.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".inc", ".h"))]
    return hs + [os.path.join(HERE, "..", "include", "mi_jpeg.h")]


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for x in extra:
        h.update(x.encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def source_hash():
    """SHA-256 over all sources, headers and compile flags of libmijpeg.so."""
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + _headers(), FLAGS)       # a missing source raises: never a silent partial library


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or _read(STAMP) != source_hash()


def build(force=False, verbose=False):
    want = source_hash()
    if not force and os.path.exists(LIB) and _read(STAMP) == want:
        return LIB
    objs = []
    hdrs = _headers()
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            raise FileNotFoundError("translation unit %s listed in build.SOURCES is missing" % path)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        # the tree hash is compiled into mij_api.hip only, so the other objects stay cached across unrelated edits
        defs = ['-DMIJ_SOURCE_HASH="%s"' % want] if src == "mij_api.hip" else []
        ohash = _digest([path] + hdrs, FLAGS + defs)
        if force or not os.path.exists(obj) or _read(obj + ".srchash") != ohash:
            cmd = [_hipcc()] + FLAGS + defs + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            with open(obj + ".srchash", "w") as f:
                f.write(ohash)
        objs.append(obj)
    link = ["g++", "-shared", "-o", LIB] + objs  # no -lamdhip64 on purpose (see module docstring)
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    with open(STAMP, "w") as f:
        f.write(want)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
