#!/usr/bin/env python3
"""Decode THROUGHPUT with several decoder handles in flight (one host thread and one stream each): the dense passes of the baseline decoder are
one round of 3.7 waves per SIMD, so two files decoded side by side share the machine. Own full-size 4:2:2 q95 file, device resident."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvjpeg_imagecompressor_amd as mij
from nvjpeg_imagecompressor_amd import sharded
W, H = 8320, 40000
dev = torch.device("cuda:0")
img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=True)
with mij.Encoder(W, H, 95, True, 1) as enc:
    enc.encode_device(img.data_ptr(), W * 3, "bgr")
    r = enc.result()
    d_jpg = sharded.device_bytes(torch, r["d_buffer"] + r["header_offset"], r["file_bytes"], dev).clone()
n = int(d_jpg.numel())
out = {}
for nthreads in (1, 2, 3):
    decs = [mij.Decoder() for _ in range(nthreads)]
    outs = [torch.empty_like(img) for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    reps = 12
    def work(k):
        for _ in range(reps):
            decs[k].decode_device_ptr(d_jpg.data_ptr(), n, outs[k].data_ptr(), W * 3, "bgr", 0, streams[k].cuda_stream)
            decs[k].sync()
    for k in range(nthreads):      # warm-up (workspaces)
        decs[k].decode_device_ptr(d_jpg.data_ptr(), n, outs[k].data_ptr(), W * 3, "bgr", 0, streams[k].cuda_stream); decs[k].sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(o, outs[0]) for o in outs)
    out["%d in flight" % nthreads] = {"ms_per_file": round(dt / (reps * nthreads) * 1e3, 3), "files": reps * nthreads, "identical_outputs": ok}
    for d in decs: d.close()
    del outs
print(json.dumps(out))
