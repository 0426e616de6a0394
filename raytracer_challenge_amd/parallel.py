"""Tile partition of one frame over the GPUs of a node and the end-of-frame gather.

Pixels are independent (reference src/image.rs:68-73), so the only exchange is one gather of finished tiles to rank 0.
The image is cut into BANDS of `BAND_ROWS` = 8 rows dealt round-robin — rank r of N owns bands r, r+N, r+2N, … — because
contiguous blocks are badly imbalanced (the mesh and its reflections sit mid-frame) and because a wave of the trace kernels
is an 8x8 pixel tile of the rank's own dense tile: with 8-row bands that tile is an 8x8 tile of the image too (single rows
interleaved by rank, the round-2 partition, spread one wave over 8·N image rows: `band_rows=1`).  The last band of the image may
be short; it is the last band of its owner's tile.  `torch.distributed` is plumbing: backend "nccl" is RCCL over xGMI on the GPUs,
"gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional

BAND_ROWS = 8


def rows_of(rank: int, world_size: int, vsize: int, band_rows: int = BAND_ROWS) -> List[int]:
    """Image rows owned by `rank`, in the order of its dense tile (include/rtc.h rtc_render_bands_device)."""
    rows: List[int] = []
    n_bands = (vsize + band_rows - 1) // band_rows
    for b in range(rank, n_bands, world_size):
        rows.extend(range(b * band_rows, min(vsize, (b + 1) * band_rows)))
    return rows


def max_rows(world_size: int, vsize: int, band_rows: int = BAND_ROWS) -> int:
    """Rows of the largest tile, padded to whole bands (every rank's gather buffer has this many)."""
    n_bands = (vsize + band_rows - 1) // band_rows
    return ((n_bands + world_size - 1) // world_size) * band_rows


class FrameGatherer:
    """Owns the gather buffers of one frame size.  `tile` is each rank's dense buffer of its own rows (float64,
    max_rows*hsize*3, padded to whole bands and to the largest rank's band count)."""

    def __init__(self, hsize: int, vsize: int, rank: int, world_size: int, device, dist=None, n_buffers: int = 2, tile_device=None,
                 band_rows: int = BAND_ROWS):
        import torch
        self.hsize, self.vsize, self.rank, self.world_size, self.dist = hsize, vsize, rank, world_size, dist
        self.band_rows = band_rows
        self.n_rows = len(rows_of(rank, world_size, vsize, band_rows))
        self.max_rows = max_rows(world_size, vsize, band_rows)
        self.max_bands = self.max_rows // band_rows
        # n_buffers tiles: frame i renders into tiles[i % n_buffers] while earlier frames are still in flight / being gathered
        # tile_device != device only in rehearsals of the multi-rank path on one GPU (tiles in HBM, gloo gather through host memory)
        self.gather_device = device
        self.tiles = [torch.zeros(self.max_rows * hsize * 3, dtype=torch.float64, device=tile_device if tile_device is not None else device)
                      for _ in range(max(1, n_buffers))]
        self.tile = self.tiles[0]
        self.gathered: Optional[List] = None
        self.image = None
        if rank == 0:
            if world_size > 1:
                # one slab [rank][band j of that rank][row in band][x][rgb]; gather_list entries are views of it, so the
                # de-interleave is ONE copy
                self.slab = torch.zeros((world_size, self.max_bands, band_rows, hsize, 3), dtype=torch.float64, device=device)
                self.gathered = [self.slab[r].view(-1) for r in range(world_size)]
                # image band r + N*j  <-  slab[r, j]: a padded (max_bands*N*band_rows)-row frame viewed as [j][r][row in band]
                self.padded = torch.zeros((self.max_bands * world_size * band_rows, hsize, 3), dtype=torch.float64, device=device)
                self.image = self.padded[:vsize]
            else:
                self.image = torch.zeros((vsize, hsize, 3), dtype=torch.float64, device=device)

    def gather(self, which: int = 0):
        """All ranks call this after tiles[which] is complete.  Rank 0 returns the assembled (vsize, hsize, 3) image, others None."""
        H, V, N = self.hsize, self.vsize, self.world_size
        tile = self.tiles[which]
        if tile.device != self.image.device if self.image is not None else False:
            tile = tile.to(self.image.device)
        elif self.image is None and str(tile.device) != str(self.gather_device):
            tile = tile.to(self.gather_device)
        if N == 1:
            self.image.view(-1)[:] = tile[: V * H * 3]
            return self.image
        self.dist.gather(tile, self.gathered, dst=0)
        if self.rank != 0:
            return None
        self.padded.view(self.max_bands, N, self.band_rows, H, 3).copy_(self.slab.permute(1, 0, 2, 3, 4))
        return self.image


class FrameRoundGatherer:
    """Frame-parallel use of N GPUs: when the caller has at least N frames queued, rank r renders WHOLE frames
    r, r + N, ... and a round of N finished frames is delivered to rank 0 with one gather (50 MB per 1080p f64 frame over xGMI, the
    same bytes per frame as the band partition).  A frame's latency stays one GPU's frame time; throughput scales with N, which the
    band partition of a ~2 ms frame does not (its per-rank share is 14 dependent launches of tens of microseconds each).
    `tiles[k]` is this rank's frame buffer k (float64, vsize*hsize*3); rank 0's `frames[r]` is the frame rank r delivered last."""

    def __init__(self, hsize: int, vsize: int, rank: int, world_size: int, device, dist=None, n_buffers: int = 2, tile_device=None):
        import torch
        self.hsize, self.vsize, self.rank, self.world_size, self.dist = hsize, vsize, rank, world_size, dist
        self.band_rows = 1
        self.n_rows = vsize
        self.gather_device = device
        self.tiles = [torch.zeros(vsize * hsize * 3, dtype=torch.float64, device=tile_device if tile_device is not None else device)
                      for _ in range(max(1, n_buffers))]
        self.tile = self.tiles[0]
        self.frames = None
        self.gathered = None
        if rank == 0:
            self.frames = torch.zeros((world_size, vsize, hsize, 3), dtype=torch.float64, device=device)
            self.gathered = [self.frames[r].view(-1) for r in range(world_size)]
        self.image = self.frames[0] if self.frames is not None else None

    def gather(self, which: int = 0):
        tile = self.tiles[which]
        if str(tile.device) != str(self.gather_device):
            tile = tile.to(self.gather_device)
        if self.world_size == 1:
            self.frames[0].view(-1)[:] = tile
            return self.frames
        self.dist.gather(tile, self.gathered, dst=0)
        return self.frames if self.rank == 0 else None
