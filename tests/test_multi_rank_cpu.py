"""The N>1 path on CPU: two processes, `gloo`, world_size 2 (and uneven 3-rank splits: a short last band, fewer bands than
ranks) run the same partition + gather + de-interleave code bench.py uses on the GPUs; every rank's tile comes from the
emulated kernel source THROUGH the band entry point of the C ABI (rtc_render_bands_device: the band -> image-row map is the
kernels' own), and the assembled frame must equal a single-rank render bit for bit."""
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


def _worker(rank, world_size, port, hsize, vsize, out_path, band_rows):
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
    from raytracer_challenge_amd.device import DeviceRenderer
    fg = FrameGatherer(hsize, vsize, rank, world_size, torch.device("cpu"), dist, band_rows=band_rows)
    rows = np.array(rows_of(rank, world_size, vsize, band_rows), dtype=np.uint64)
    assert len(rows) == fg.n_rows
    dr = DeviceRenderer(be, nw, cam, _cpu_standin=True)
    dr.render_rows(5, rank, world_size, fg.n_rows, fg.tile, band_rows=band_rows)
    if len(rows):  # the same pixels named one by one: the dense tile is rows_of()'s rows in order
        idx = (rows[:, None] * hsize + np.arange(hsize, dtype=np.uint64)[None, :]).reshape(-1)
        rgb, _ = be.render(nw, cam, 5, idx)
        assert np.array_equal(fg.tile[: rgb.size].numpy(), rgb.reshape(-1))
    img = fg.gather()
    if rank == 0:
        np.save(out_path, img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size,vsize,band_rows", [(2, 24, 8), (3, 25, 8), (3, 12, 8), (3, 25, 1), (2, 23, 3)])
def test_band_gather_matches_single_rank(tmp_path, world_size, vsize, band_rows):
    from emu_lib import emu
    from raytracer_challenge_amd import scenes
    hsize = 40
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world_size, _free_port(), hsize, vsize, out, band_rows), nprocs=world_size, join=True)
    cam, world = scenes.chapter11_title(hsize, vsize)
    be = emu()
    full, _ = be.render(be.build_world(world), cam, 5)
    assert np.array_equal(np.load(out).reshape(-1, 3), full)


def test_band_rows_owned_matches_the_python_partition():
    """rtc_band_rows_owned (host arithmetic of the product library: callable without a GPU) against parallel.rows_of."""
    import ctypes as C
    from raytracer_challenge_amd.parallel import rows_of, max_rows
    lib = C.CDLL(os.path.join(ROOT, "raytracer_challenge_amd", "csrc", "librtc_amd.so"))
    lib.rtc_band_rows_owned.restype = C.c_uint64
    lib.rtc_band_rows_owned.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    for vsize in (1, 7, 8, 9, 25, 1080, 2160, 2161):
        for band in (1, 3, 8, 32):
            for n in (1, 2, 3, 4, 8):
                owned = [len(rows_of(k, n, vsize, band)) for k in range(n)]
                assert sum(owned) == vsize
                assert sorted(sum((rows_of(k, n, vsize, band) for k in range(n)), [])) == list(range(vsize))
                assert max(owned) <= max_rows(n, vsize, band)
                for k in range(n):
                    assert lib.rtc_band_rows_owned(vsize, band, k, n) == owned[k], (vsize, band, n, k)
