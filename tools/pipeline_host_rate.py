"""How fast can ONE rank drive sharded.DevicePipeline? A single rank over RCCL (its own root) with a strip the size one of
eight ranks gets at the full size (8320 x 5000 of 8320 x 40000): wall time per image over 400 images, and the host's share of it
(time spent inside step() alone). At 8 GPUs a step must stay below ~0.2 ms for 6x; the put over xGMI is the only part missing here."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import nvjpeg_imagecompressor_amd as mij
from nvjpeg_imagecompressor_amd import sharded

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
W, H = 8320, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
COMMS = sys.argv[2] if len(sys.argv) > 2 else "ordered"       # "ordered" (one communicator, the default) | "per-slot"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
whole, r0, r1 = sharded.strip_rows(W, H, 95, True, 1, 0, 1)
encs = [sharded.make_hip_strip_encoder(torch, W, H, 95, True, 1, 0, 1, 0, "bgr") for _ in range(sharded.DEPTH)]
img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
mij.synth_image_device(img.data_ptr(), W, 0, H, W * 3, bgr=True)
torch.cuda.synchronize()
strips = [sharded.HipStripEncoder(torch, e, img, "bgr") for e in encs]
targets = sharded.open_file_targets(torch, dist, strips, 0, 1, 0, whole)
pipe = sharded.DevicePipeline(torch, dist, strips, targets, True, device=dev, comms=COMMS)
for _ in range(20):
    pipe.step()
pipe.flush()
torch.cuda.synchronize()
N = 400
host = 0.0
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    pipe.step()
    host += time.perf_counter() - a
issued = time.perf_counter() - t0
pipe.flush()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(json.dumps({"strip": "%dx%d" % (W, H), "comms": COMMS, "images": N, "ms_per_image_wall": round(wall / N * 1e3, 4),
                  "ms_per_image_host_in_step": round(host / N * 1e3, 4), "ms_per_image_until_all_issued": round(issued / N * 1e3, 4)}))
for e in encs:
    e.close()
dist.destroy_process_group()
