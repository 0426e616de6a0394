"""The N>1 path on CPU: two processes, `gloo`, world_size 2 (and an uneven 3-rank split) run the same partition +
gather + de-interleave code bench.py uses on the GPUs; tiles come from the emulated kernel source, and the assembled
frame must equal a single-rank render bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, hsize, vsize, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from emu_lib import emu
    from raytracer_challenge_amd import scenes
    from raytracer_challenge_amd.parallel import FrameGatherer, rows_of
    cam, world = scenes.chapter11_title(hsize, vsize)
    be = emu()
    nw = be.build_world(world)
    fg = FrameGatherer(hsize, vsize, rank, world_size, torch.device("cpu"), dist)
    rows = np.array(list(rows_of(rank, world_size, vsize)), dtype=np.uint64)
    idx = (rows[:, None] * hsize + np.arange(hsize, dtype=np.uint64)[None, :]).reshape(-1)
    rgb, _ = be.render(nw, cam, 5, idx)
    fg.tile[: rgb.size] = torch.from_numpy(rgb.reshape(-1))
    img = fg.gather()
    if rank == 0:
        np.save(out_path, img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size,vsize", [(2, 24), (3, 25)])
def test_row_interleaved_gather_matches_single_rank(tmp_path, world_size, vsize):
    from emu_lib import emu
    from raytracer_challenge_amd import scenes
    hsize = 40
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world_size, _free_port(), hsize, vsize, out), nprocs=world_size, join=True)
    cam, world = scenes.chapter11_title(hsize, vsize)
    be = emu()
    full, _ = be.render(be.build_world(world), cam, 5)
    assert np.array_equal(np.load(out).reshape(-1, 3), full)
