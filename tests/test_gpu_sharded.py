"""Strip sharding with the HIP strip encoder: two ranks share the one GPU of the test box (gloo carries the collectives;
RCCL refuses two ranks on one device), each encodes its strip, rank 0 gathers. Must equal the oracle's one-shot file."""
import os
import socket
import sys
import zlib

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, W, H, css, optimize, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nvjpeg_imagecompressor_amd as mij
    from nvjpeg_imagecompressor_amd import sharded
    torch.cuda.set_device(0)
    enc = sharded.make_hip_strip_encoder(torch, W, H, 95, optimize, css, rank, world, 0, "rgb")
    g = enc.geometry
    d_img = torch.empty((g["strip_rows"], W, 3), dtype=torch.uint8, device="cuda:0")
    mij.synth_image_device(d_img.data_ptr(), W, g["strip_y0"], g["strip_rows"], W * 3, bgr=False)
    torch.cuda.synchronize()
    strip = sharded.HipStripEncoder(torch, enc, d_img, "rgb")
    out = None
    for _ in range(2):   # twice: buffers are reused across steps
        out = sharded.encode_step(torch, dist, strip, optimize, {})
    if rank == 0:
        open(out_path, "wb").write(out.cpu().numpy().tobytes())
        open(out_path + ".ri", "w").write(str(g["restart_interval"]))
    dist.barrier()
    enc.close()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("css,optimize", [(1, True), (2, True), (0, False)])
def test_hip_strips_equal_one_shot(oracle, tmp_path, world, css, optimize):
    W, H = 2080, 1000    # 1000 rows: not a multiple of 16, the last strip owns the bottom edge
    out = tmp_path / "sharded.jpg"
    mp.spawn(_worker, args=(world, _free_port(), W, H, css, optimize, str(out)), nprocs=world, join=True)
    ri = int(open(str(out) + ".ri").read())
    want = oracle.encode(oracle.synth_rgb(W, H), 95, css, optimize, ri)
    got = out.read_bytes()
    assert len(got) == len(want) and zlib.crc32(got) == zlib.crc32(want)


def test_hip_strips_five_ranks_uneven(oracle, tmp_path):
    """Five ranks (the most the test box lets share its GPU next to the test process), strips of unequal height, bottom
    edge inside the last strip, 4:2:0 so that an MCU row is 16 pixels."""
    W, H, css, world = 4160, 2504, 2, 5
    out = tmp_path / "sharded5.jpg"
    mp.spawn(_worker, args=(world, _free_port(), W, H, css, True, str(out)), nprocs=world, join=True)
    ri = int(open(str(out) + ".ri").read())
    want = oracle.encode(oracle.synth_rgb(W, H), 95, css, True, ri)
    got = out.read_bytes()
    assert len(got) == len(want) and zlib.crc32(got) == zlib.crc32(want)
