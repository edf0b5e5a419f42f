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

from .bindings import Partition

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
