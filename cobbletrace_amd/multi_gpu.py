"""Row-tile sharding of one frame over the GPUs of a node (one process per GPU).

The reference shards a frame over worker threads by contiguous row bands
(raythread.cpp:574-588) and the workers share the framebuffer.  Across GPUs the same rows
are cut into tiles of `tile_rows` rows dealt round-robin to the ranks (interleaved, because
the geometry sits mid-image), every rank traces its tiles with the scene replicated in its
own HBM, and ONE collective per frame -- an RCCL gather of the uint32 tile buffers to rank 0
over xGMI -- assembles the image; no other data-path communication exists.

torch is used for what it is here for: device buffers, streams and torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" for the CPU rehearsal of the plumbing).
"""
import ctypes as C

import torch
import torch.distributed as dist

from .bindings import GpuContext, Partition

DEFAULT_TILE_ROWS = 8   # one wave tile (8x8 pixels) high


def partition_of(rank, world, tile_rows=DEFAULT_TILE_ROWS):
    return Partition(tile_rows, rank, world)


def gather_to_root(local, world, rank, root=0):
    """All ranks' equally sized tile buffers, back to back on `root` (None elsewhere)."""
    if world == 1 and not dist.is_initialized():
        return local
    if rank == root:
        everything = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        pieces = list(everything.split(local.numel()))
        dist.gather(local, gather_list=pieces, dst=root)
        return everything
    dist.gather(local, gather_list=None, dst=root)
    return None


class ShardedRenderer:
    """Per-rank: trace my tiles -> gather -> (rank 0) assemble into a device framebuffer."""

    def __init__(self, scene_desc, rank, world, device_index, tile_rows=DEFAULT_TILE_ROWS):
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = GpuContext(device_index)
        self.ctx.upload(scene_desc)
        self.part = partition_of(rank, world, tile_rows)
        self._buffers = {}

    def _buffers_for(self, frame):
        key = (frame.width, frame.height, frame.num_threads)
        if key not in self._buffers:
            n = self.ctx.partition_pixels(frame, self.part)
            tiles = torch.zeros(n, dtype=torch.int32, device=self.device)
            fb = torch.zeros(frame.height * frame.width, dtype=torch.int32, device=self.device) if self.rank == 0 else None
            self._buffers[key] = (tiles, fb)
        return self._buffers[key]

    def trace(self, frame):
        """Launch this rank's trace kernel on torch's current stream; returns the tile buffer."""
        tiles, _ = self._buffers_for(frame)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.render_partition(frame, self.part, tiles.data_ptr(), stream)
        return tiles

    def frame(self, frame):
        """One whole frame; rank 0 gets the device framebuffer (int32 view of 0x00BBGGRR), others None."""
        tiles = self.trace(frame)
        gathered = gather_to_root(tiles, self.world, self.rank)
        if self.rank != 0:
            return None
        _, fb = self._buffers_for(frame)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.assemble(frame, self.part, gathered.data_ptr(), fb.data_ptr(), stream)
        return fb

    def close(self):
        self.ctx.close()
