"""Regenerates tests/golden/big_8320x40000_q95.json: length + CRC32 of the CPU oracle's output for BASELINE.json's
full-size configurations (the files themselves are 150-310 MB, so only their fingerprints are committed).
Runs on the CPU only, about 1-2 minutes per case:  python tests/make_golden_big.py [case ...]"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402

W, H = 8320, 40000
# (key, css, restart interval, optimise, progressive): the intervals the encoder's AUTO rule picks (64 / 32 MCUs baseline, 640
# progressive) and round 1's choices (104 / 52 / 80), which the tests keep encoding with an explicit interval
CASES = [("css1_ri64_opt", 1, 64, True, False), ("css0_ri64_opt", 0, 64, True, False), ("css2_ri32_opt", 2, 32, True, False),
         ("css3_ri64_opt", 3, 64, True, False), ("css4_ri32_opt", 4, 32, True, False), ("css1_ri64_fix", 1, 64, False, False),
         ("css1_ri104_opt", 1, 104, True, False), ("css0_ri104_opt", 0, 104, True, False), ("css2_ri52_opt", 2, 52, True, False),
         ("css3_ri80_opt", 3, 80, True, False), ("css4_ri52_opt", 4, 52, True, False), ("css1_ri104_fix", 1, 104, False, False),
         ("css1_ri104_progressive", 1, 104, True, True), ("css1_ri640_progressive", 1, 640, True, True),
         ("css1_ri520_progressive", 1, 520, True, True)]      # 520 = the MCU row: the interval of a SHARDED progressive encode (sharded.progressive_strip_interval)


def main():
    path = os.path.join(HERE, "golden", "big_8320x40000_q95.json")
    gold = json.load(open(path)) if os.path.exists(path) else {"cases": {}}
    img = np.empty((H, W, 3), np.uint8)
    for y in range(0, H, 4000):
        img[y:y + 4000] = O.synth_rgb(W, H, y0=y, rows=4000)
    gold["synthetic_crc32"] = "%08x" % zlib.crc32(img.tobytes())
    want = set(sys.argv[1:])
    for key, css, ri, opt, prog in CASES:
        if want and key not in want:
            continue
        j = O.encode_progressive(img, 95, css, ri) if prog else O.encode(img, 95, css, opt, ri)
        gold["cases"][key] = {"len": len(j), "crc32": "%08x" % zlib.crc32(j)}
        print(key, gold["cases"][key], flush=True)
    json.dump(gold, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
