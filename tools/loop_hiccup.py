#!/usr/bin/env python3
"""Where do the ~8 ms stalls of the encode loop come from? Host timestamps around every library call of 400 images of the overlap
loop; prints the calls that took more than 1 ms and the completion-time gaps above 2 ms."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvjpeg_imagecompressor_amd as mij
from nvjpeg_imagecompressor_amd import sharded

W, H = 8320, 40000
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
encs = [sharded.make_hip_strip_encoder(torch, W, H, 95, True, "422", 0, 1, 0, "bgr") for _ in range(3)]
main = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
d_img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(d_img.data_ptr(), W, 0, H, W * 3, bgr=True, stream=main)
strips = [sharded.HipStripEncoder(torch, e, d_img, "bgr", shared_statistics=False) for e in encs]
torch.cuda.synchronize()
gc.disable()
ev = []
q = []
pc = time.perf_counter
T0 = pc()
for i in range(N):
    st = strips[i % 3]
    a = pc(); st.enc.transform(st.d_img.data_ptr(), st.pitch, st.fmt, 0, main)
    b = pc(); st.enc.tables(side.cuda_stream)
    c = pc(); st.enc.entropy(side.cuda_stream)
    d = pc()
    q.append(st)
    e = d
    if len(q) >= 3:
        q.pop(0).finish_whole()
        e = pc()
    ev.append((i, a - T0, b - a, c - b, d - c, e - d))
torch.cuda.synchronize()
print("total %.1f ms for %d images = %.4f ms/image" % ((pc() - T0) * 1e3, N, (pc() - T0) * 1e3 / N))
prev = None
for (i, t, xf, tb, en, fin) in ev:
    if xf > 1e-3 or tb > 1e-3 or en > 1e-3 or fin > 2.5e-3:
        print("image %4d at %8.2f ms: transform %.3f tables %.3f entropy %.3f finish %.3f ms" % (i, t * 1e3, xf * 1e3, tb * 1e3, en * 1e3, fin * 1e3))
