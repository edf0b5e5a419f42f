import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
scene = ca.HostScene("/root/repo/assets/scene_import_dragon.json")
frame = scene.frame(1920, 1080, rng_mode=ca.RNG_HASH, rng_seed=0)
for world in (1, 2, 4, 8):
    part = ca.Partition(8, 0, world)
    ctx = ca.GpuContext(0); ctx.upload(scene.desc, build_on_device=True)
    n = ctx.partition_pixels(frame, part)
    stream = torch.cuda.Stream()
    tiles = torch.zeros(n * world, dtype=torch.int32, device="cuda")
    fb = torch.zeros(frame.height * frame.width, dtype=torch.int32, device="cuda")
    host = torch.zeros(frame.height * frame.width, dtype=torch.int32).pin_memory()
    def one():
        with torch.cuda.stream(stream):
            ctx.render_partition(frame, part, tiles.data_ptr(), stream.cuda_stream)
            ctx.assemble(frame, part, tiles.data_ptr(), fb.data_ptr(), stream.cuda_stream)
            ctx.download_frame(frame, fb.data_ptr(), host.data_ptr(), stream.cuda_stream)
    for _ in range(5): one()
    torch.cuda.synchronize()
    ctx.kernel_timing(True); ctx.read_kernel_times(); ctx.read_walk_work()
    N = 40
    t0 = time.perf_counter()
    for _ in range(N): one()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e3
    t = ctx.read_kernel_times(); w = ctx.read_walk_work()
    per = [t.total_ms[i] / max(1, t.launches[i]) for i in range(5)]
    print("world %d: one frame at a time %.3f ms wall; kernels %s sum %.3f; packets %d + %d" % (world, wall, " ".join("%.3f" % v for v in per), sum(per), w.packets[0] / N, w.packets[1] / N), flush=True)
    ctx.close()
