#!/usr/bin/env python3
"""How long does the device take to reach its steady-state rate when the encode loop starts? ms per image in chunks of 5 images,
for the overlap loop and the one-stream loop, from idle and right behind a continuous one-stream pass."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvjpeg_imagecompressor_amd as mij
from nvjpeg_imagecompressor_amd import sharded

W, H = 8320, 40000
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
encs = [sharded.make_hip_strip_encoder(torch, W, H, 95, True, "422", 0, 1, 0, "bgr") for _ in range(3)]
main = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
d_img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(d_img.data_ptr(), W, 0, H, W * 3, bgr=True, stream=main)
strips = [sharded.HipStripEncoder(torch, e, d_img, "bgr", shared_statistics=False) for e in encs]
torch.cuda.synchronize()


def run(loop, n, label, chunk=5):
    q, t = [], []
    t0 = time.perf_counter()
    for i in range(n):
        st = strips[i % 3]
        if loop == "overlap":
            st.enc.transform(st.d_img.data_ptr(), st.pitch, st.fmt, 0, main)
            st.enc.tables(side.cuda_stream)
            st.enc.entropy(side.cuda_stream)
        else:
            st.issue_whole(main)
        q.append(st)
        if len(q) >= 3:
            q.pop(0).finish_whole()
            t.append(time.perf_counter())
    while q:
        q.pop(0).finish_whole()
        t.append(time.perf_counter())
    torch.cuda.synchronize()
    per = [(t[i + chunk] - t[i]) / chunk * 1e3 for i in range(0, len(t) - chunk, chunk)]
    print("%-44s" % label, " ".join("%.3f" % p for p in per), flush=True)
    return time.perf_counter() - t0


def clk(tag):
    c = mij.clock_probe_device(1024, main)
    print("   clock %-30s valu %.0f counter %.0f" % (tag, c["valu_mhz"], c["counter_mhz"]), flush=True)


for rep in range(2):
    time.sleep(0.5)
    clk("idle")
    run("overlap", 150, "overlap from idle")
    clk("after overlap 150")
    time.sleep(0.5)
    run("one", 150, "one-stream from idle")
    clk("after one-stream 150")
    time.sleep(0.5)
    run("one", 25, "one-stream 25 (conditioning)")
    run("overlap", 60, "overlap right behind it")
    clk("after")
    time.sleep(0.5)
    run("overlap", 6, "overlap 6 (warm-up)", chunk=2)
    torch.cuda.synchronize()
    clk("behind 6 overlap")
    run("overlap", 40, "overlap 40 behind 6 + probe")
    clk("after")
