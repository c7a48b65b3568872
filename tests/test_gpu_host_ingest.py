"""mij_encode_host: the pipelined upload (ranges of MCU rows, stage A overlapped with the next upload) must produce the
same file as the oracle from pageable and from page-locked memory, for interleaved and planar input, with several
upload ranges and with a partial last MCU row (reference path: ImageCompressorImpl.cu:269-294)."""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,css,opt", [(8320, 4000, 1, True), (8320, 2999, 2, True), (4100, 6001, 4, False), (9000, 1300, 0, True)])
def test_multi_range_upload_matches_oracle(mij, oracle, w, h, css, opt):
    img = oracle.synth_rgb(w, h)                       # 37-100 MB: 2-4 upload ranges of 32 MiB
    with mij.Encoder(w, h, 95, opt, css) as enc:
        ri = enc.geometry["restart_interval"]
        want = oracle.encode(img, 95, css, opt, ri)
        assert enc.encode_host(img, "rgb") == want
        pinned = mij.pinned_empty(img.shape)
        pinned[...] = img
        assert bytes(enc.encode_host(pinned, "rgb", as_view=True)) == want
        planes = np.ascontiguousarray(img.transpose(2, 0, 1))
        assert enc.encode_host(planes, "rgb_planar") == want


def test_full_size_from_pinned_host_memory(mij, oracle):
    """BASELINE config from host memory (31 upload ranges): the committed golden length / CRC."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "big_8320x40000_q95.json")))["cases"]
    W, H = 8320, 40000
    gold = gold["css1_ri%d_opt" % mij.geometry_query(W, H, 95, True, 1)["restart_interval"]]
    img = mij.pinned_empty((H, W, 3))
    for y in range(0, H, 4000):
        img[y:y + 4000] = oracle.synth_rgb(W, H, y0=y, rows=4000)
    with mij.Encoder(W, H, 95, True, 1) as enc:
        out = enc.encode_host(img, "rgb", as_view=True)
        assert len(out) == gold["len"] and "%08x" % zlib.crc32(out) == gold["crc32"]
