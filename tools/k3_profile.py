#!/usr/bin/env python3
"""Where k_build_tables spends its time: builds libmijpeg.so with -DMIJ_K3_PROFILE (cycle stamps of the AC-luma wave in the
unused front of the header area), encodes the bench image once and prints the phases.
    python tools/k3_profile.py build     (no GPU needed)
    python tools/k3_profile.py run       (on the GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nvjpeg_imagecompressor_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants", "k3prof")


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in ("mij_kernels.hip", "mij_api.hip", "mij_decode_api.hip"):
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-DMIJ_FAST_BUILD", "-DMIJ_K3_PROFILE", "-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-o", os.path.join(OUT, "libmijpeg.so")] + objs)


def run():
    os.environ["MIJ_LIB_PATH"] = os.path.join(OUT, "libmijpeg.so")
    sys.path.insert(0, ROOT)
    import torch
    import nvjpeg_imagecompressor_amd as mij
    W, H = 8320, 40000
    d = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    mij.synth_image_device(d.data_ptr(), W, 0, H, W * 3, bgr=False)
    with mij.Encoder(W, H, 95, True, 1) as enc:
        for _ in range(3):
            enc.encode_device(d.data_ptr(), W * 3, "rgb")
            enc.result()
        p, _, _ = enc.output_buffer()
        st = torch.empty(8, dtype=torch.int64)
        ctypes.CDLL(None)
        torch.cuda.synchronize()
        buf = torch.empty(64, dtype=torch.uint8, device="cuda:0")
        # the stamps sit at the very front of the output buffer
        import ctypes as C
        hip = C.CDLL(torch.__file__.replace("__init__.py", "lib/libamdhip64.so"))
        host = (C.c_ulonglong * 8)()
        rc = hip.hipMemcpy(host, C.c_void_p(p), 64, 2)
        assert rc == 0, rc
        names = ["load+compact", "registers", "merge loop", "sync", "count + K.3 limit", "huffval sort", "codes + LUT", "header"]
        t = list(host)
        for i in range(1, 8):
            print("%-20s %8d cycles" % (names[i - 1] if i - 1 < len(names) else i, t[i] - t[i - 1]))
        print("total %d cycles (s_memtime / shader clock)" % (t[7] - t[0]))


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
