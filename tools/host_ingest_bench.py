"""PCIe-inclusive encode rate: host-resident BGR image in, host-resident JFIF out (mij_encode_host), i.e. the whole of
the reference's compress() including the marshalling its README leaves out of the timing (ImageCompressorImpl.cu:272-287).
Prints one JSON line with the rate from pageable and from page-locked source memory. Never the bench.py `value`."""
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch

import nvjpeg_imagecompressor_amd as mij

W, H = 8320, 40000


def timed(enc, img, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = enc.encode_host(img, "bgr", as_view=True)
        best = min(best, time.perf_counter() - t0)
    return best, len(out)


def main():
    pinned = mij.pinned_empty((H, W, 3))
    d = torch.empty((4000, W, 3), dtype=torch.uint8, device="cuda:0")
    for y in range(0, H, 4000):     # the library's device generator (SURVEY 8d synthetic image), BGR
        mij.synth_image_device(d.data_ptr(), W, y, 4000, W * 3, bgr=True)
        torch.cuda.synchronize()
        pinned[y:y + 4000] = d.cpu().numpy()
    del d
    pageable = np.array(pinned)
    enc = mij.Encoder(W, H, 95, True, 1)
    enc.encode_host(pinned, "bgr", as_view=True)      # warm-up: allocations, first-touch
    tp, n = timed(enc, pinned)
    tg, _ = timed(enc, pageable)
    mb = W * H * 3 / 1e6
    print(json.dumps({"workload": "8320x40000 BGR8 host -> JFIF host, q95 4:2:2 optimised", "jpeg_bytes": n,
                      "pinned": {"ms": round(tp * 1e3, 2), "Mpixels/s": round(W * H / 1e6 / tp, 1), "GB/s_in": round(mb / 1e3 / tp, 2)},
                      "pageable": {"ms": round(tg * 1e3, 2), "Mpixels/s": round(W * H / 1e6 / tg, 1), "GB/s_in": round(mb / 1e3 / tg, 2)}}))
    enc.close()


if __name__ == "__main__":
    main()
