"""The N > 1 path on CPU: two gloo ranks shard a frame by interleaved row tiles exactly as
bench.py does on N GPUs (cobbletrace_amd.multi_gpu), gather the tile buffers to rank 0 and
assemble them.  The per-rank tracing stands in for the HIP kernel by using the oracle (this is a
test); the partition arithmetic (ctgpu_partition_pixels), the gather plumbing and the assemble
(cth_assemble_tiles, compiled from the same include/ct_tiles.h as the ct_assemble kernel) are the product's."""
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
    port = s.getsockname()[1]
    s.close()
    return port


def _local_tiles(frame, part, scene_desc, oracle, n_pixels):
    """What ctgpu_render_partition leaves in a rank's tile buffer, produced with the oracle:
    local row r of this rank = frame row ((r // tile_rows) * count + index) * tile_rows + r % tile_rows."""
    half = frame.height // 2
    square = 2 * half
    rows = frame.num_threads * (frame.height // frame.num_threads)
    full, _ = oracle.render(scene_desc, frame, os_threads=2)
    local_rows = n_pixels // square
    tiles = np.zeros((local_rows, square), dtype=np.uint32)
    for lr in range(local_rows):
        yi = ((lr // part.tile_rows) * part.part_count + part.part_index) * part.tile_rows + lr % part.tile_rows
        row = frame.height // 2 - (yi - half)
        if yi < rows and 0 <= row < frame.height:
            tiles[lr] = full[row, frame.width // 2 - half: frame.width // 2 + half]
    return tiles


def _assemble(frame, tile_rows, world, gathered):
    """The product's own host-side assemble (libcthost.so cth_assemble_tiles; the same include/ct_tiles.h
    arithmetic the ct_assemble kernel is compiled from)."""
    import ctypes as C
    import cobbletrace_amd as ca
    fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
    gathered = np.ascontiguousarray(gathered, dtype=np.uint32)
    assert ca.host_library().cth_assemble_tiles(C.byref(frame), tile_rows, world, gathered.ctypes.data, fb.ctypes.data) == 0
    return fb


def _worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cobbletrace_amd as ca
    from cobbletrace_amd.multi_gpu import gather_to_root, partition_of
    from oracle import oracle
    scene = ca.HostScene(os.path.join(ROOT, "assets", "scene_file_cube.json"))
    ok = True
    for tile_rows, (w, h, t) in ((8, (160, 96, 4)), (4, (100, 50, 5))):
        frame = scene.frame(w, h, threads=t, supersampling=False)
        part = partition_of(rank, world, tile_rows)
        n = ca.GpuContext.partition_pixels(None, frame, part) if False else None
        import ctypes as C
        cnt = C.c_uint64()
        assert ca.gpu_library().ctgpu_partition_pixels(C.byref(frame), C.byref(part), C.byref(cnt)) == 0
        tiles = _local_tiles(frame, part, scene.desc, oracle, cnt.value)
        local = torch.from_numpy(tiles.view(np.int32).ravel().copy())
        gathered = gather_to_root(local, world, rank)
        # timing plumbing of bench.py: barrier + max-over-ranks
        dist.barrier()
        tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ok = ok and tmax.item() == world
        if rank == 0:
            fb = _assemble(frame, tile_rows, world, gathered.numpy().view(np.uint32))
            want, _ = oracle.render(scene.desc, frame, os_threads=2)
            ok = ok and bool((fb == want).all())
        else:
            ok = ok and gathered is None
    if rank == 0:
        open(result_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_frame_over_gloo(world, tmp_path, native_build):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), str(result)), nprocs=world, join=True)
    assert result.read_text() == "ok"
