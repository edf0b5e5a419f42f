#!/usr/bin/env python3
"""bench.py -- Mrays/s of the CobbleTrace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is ONE FRAME of BASELINE.json's headline workload: scene_import_dragon.json at
1920x1080 with the scene file's own settings (32 row bands, supersampling on = 16 jittered
samples per pixel, subsampling off), jitter from the counter-based generator (the mode in
which GPU and CPU pixels are comparable).  The Stanford dragon itself is not
distributed with the reference; the mesh is the repository's deterministic 871,200-triangle
stand-in (tools/make_dragon_standin.py) -- "data" says so.

For N > 1 (launched by torch.distributed.run, one rank per GPU) the frame's row tiles are
dealt round-robin to the ranks and assembled with one RCCL gather; the total work per step
is fixed, so scaling is "strong".  value = rays of all ranks / wall time (max over ranks),
everything resident in HBM; the host-framebuffer copy is reported separately.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
# algorithmic bytes, reference layout: node 64 B, triangle test 4 B index + 72 B
# vertices, shaded hit 12 B material + 72 B vertices, pixel 4 B
B_NODE, B_TRI, B_HIT, B_PIXEL = 64, 76, 84, 4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "bunny", "import", "cube"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--supersampling", type=int, default=-1, help="-1: scene file's setting")
    ap.add_argument("--inflight", type=int, default=3, help="frames in flight (1 = strictly serial frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-bands", type=int, default=16, help="row bands (of T) the CPU baseline samples")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: run the N-rank code path with every rank on GPU 0 and the gather "
                         "staged through gloo (RCCL refuses two ranks on one device); numbers are meaningless")
    return ap.parse_args()


SCENE_FILES = {"dragon": "scene_import_dragon.json", "bunny": "scene_import_bunny.json",
               "import": "scene_import.json", "cube": "scene_file_cube.json"}


def algorithmic_bytes(st, pixels):
    hits = st.rays_shadow // 2 if st.rays_shadow else 0
    return st.node_visits * B_NODE + st.triangle_tests * B_TRI + hits * B_HIT + pixels * B_PIXEL


def cpu_baseline(scene, frame, n_bands):
    """The oracle -- a plain-C port of the reference's Boss/Worker path, one worker thread per
    row band like raythread.cpp:596-618 -- timed on this box's host cores on a bounded sample:
    n_bands of the frame's T bands spread evenly over the image, all running concurrently."""
    from oracle import oracle           # the checker, used here only as the timed CPU baseline
    import numpy as np
    cores = os.cpu_count() or 1
    T = frame.num_threads
    n_bands = max(1, min(n_bands, T))
    stride = max(1, T // n_bands)
    first = (T - ((n_bands - 1) * stride + 1)) // 2
    threads = min(cores, n_bands)
    fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
    t0 = time.perf_counter()
    _, st = oracle.render(scene.desc, frame, os_threads=threads, band_first=first, band_count=n_bands,
                          band_stride=stride, fb=fb)
    seconds = time.perf_counter() - t0
    picks = [first + i * stride for i in range(n_bands)]
    return st.rays / seconds / 1e6, threads, picks, st.rays, seconds


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import cobbletrace_amd as ca
    from cobbletrace_amd.multi_gpu import ShardedRenderer
    import __graft_entry__ as entry

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # every rank needs the assets and the native libraries; rank 0 prepares, the others wait
    if rank == 0:
        libs = [os.path.join(ROOT, "cobbletrace_amd", "csrc", "libctgpu.so"),
                os.path.join(ROOT, "cobbletrace_amd", "host", "libcthost.so"),
                os.path.join(ROOT, "oracle", "libctoracle.so")]
        if not all(os.path.exists(p) for p in libs):
            entry.build()
        entry.prepare_assets(need_dragon=(args.scene == "dragon"))
    if world > 1:
        dist.barrier()

    scene = ca.HostScene(os.path.join(ROOT, "assets", SCENE_FILES[args.scene]))
    ss = None if args.supersampling < 0 else bool(args.supersampling)
    frame = scene.frame(args.width, args.height, supersampling=ss, rng_mode=ca.RNG_HASH, rng_seed=0)
    is_ss = bool(frame.flags & ca.F_SUPERSAMPLE)

    # Frames are double-buffered (--inflight 2): frame i+1 is traced on a second stream / context
    # while frame i's kernels drain their tail and its tiles are gathered and assembled, the way a
    # renderer producing a stream of frames runs.  Each context owns its scene copy and scratch arrays.
    inflight = max(1, args.inflight)
    # Throughput tuning through the library's documented knobs (read at context creation): with several
    # frames in flight and only a slice of the frame per rank, a smaller persistent grid per context
    # and 128-ray-id chunks beat the defaults, which are tuned for one frame's latency
    # (tools/overlap.py: 8-way share 0.57 -> 0.50 ms, while a quarter of the frame is better off with the
    # defaults: 0.90 vs 0.94 ms).
    if inflight > 1 and world >= 8:
        os.environ.setdefault("CTGPU_BLOCKS_PER_CU", "12")
        os.environ.setdefault("CTGPU_CHUNK", "128")
    renderers = [ShardedRenderer(scene.desc, rank, world, local_rank) for _ in range(inflight)]
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    renderer = renderers[0]

    def gather(tiles):
        if world == 1:
            return tiles
        if args.rehearse_on_one_gpu:                      # gloo cannot move device memory: stage through the host
            torch.cuda.current_stream().synchronize()
            staged = ca.multi_gpu.gather_to_root(tiles.cpu(), world, rank)
            return staged.cuda() if staged is not None else None
        return ca.multi_gpu.gather_to_root(tiles, world, rank)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_frame(i, events=None):
        """Issue frame i: trace my tiles, gather to rank 0, assemble there (all asynchronous)."""
        r, s_ = renderers[i % inflight], streams[i % inflight]
        with torch.cuda.stream(s_):
            if events:
                events[0].record(s_)
            tiles = r.trace(frame)
            if events:
                events[1].record(s_)
            gathered = gather(tiles)
            if rank == 0:
                _, fb = r._buffers_for(frame)
                r.ctx.assemble(frame, r.part, gathered.data_ptr(), fb.data_ptr(), s_.cuda_stream)
                return fb
        return None

    for i in range(max(args.warmup, inflight)):
        one_frame(i)
    sync_all()
    for r in renderers:
        r.ctx.read_counters()             # reset the ray counters

    # ---- timed region: exactly K frames
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_frame(i)
    sync_all()
    elapsed = time.perf_counter() - t0

    counted = [r.ctx.read_counters() for r in renderers]
    my_rays = float(sum(c.rays for c in counted))
    totals = torch.tensor([my_rays, elapsed], dtype=torch.float64,
                          device="cpu" if args.rehearse_on_one_gpu else "cuda")
    if world > 1:
        rays_all = totals[:1].clone()
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        time_max = totals[1:].clone()
        dist.all_reduce(time_max, op=dist.ReduceOp.MAX)
        total_rays, elapsed = float(rays_all.item()), float(time_max.item())
    else:
        total_rays = my_rays
    rays_per_frame = total_rays / args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / elapsed / 1e6

    # ---- un-overlapped passes (outside the timed region): kernel time for the roofline, one frame's
    # latency, and the frame -> host framebuffer hand-off (PCIe included; never `value`)
    serial = 5
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(serial)]
    sync_all()
    t0 = time.perf_counter()
    for k in range(serial):
        one_frame(0, ev[k])
        sync_all()
    latency_ms = (time.perf_counter() - t0) / serial * 1e3
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / serial
    rank0_rays_per_frame = renderers[0].ctx.read_counters().rays / serial
    host_fb = torch.empty(frame.height * frame.width, dtype=torch.int32).pin_memory() if rank == 0 else None
    sync_all()
    t0 = time.perf_counter()
    fb = one_frame(0)
    if rank == 0:
        with torch.cuda.stream(streams[0]):
            host_fb.copy_(fb, non_blocking=True)
    torch.cuda.synchronize()
    to_host_ms = (time.perf_counter() - t0) * 1e3

    out = None
    if rank == 0:
        # ---- roofline of the frame trace (ct_walk_packet dominates): algorithmic bytes of ONE frame
        work = renderer.ctx.count_work(frame)           # whole frame, instrumented variant
        pixels = int(work.rays_primary // (16 if is_ss else 1))
        frame_bytes = algorithmic_bytes(work, pixels)
        launch_bytes = frame_bytes * rank0_rays_per_frame / max(work.rays, 1)     # rank 0's share of the frame
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per frame from the PMC passes of the same command (tools/collect_profiles.sh ->
        # profiles/*_traffic.json); PMC counters cannot be collected from inside this process
        traffic, traffic_src = None, None
        if world == 1 and args.scene == "dragon" and args.width == 1920 and args.height == 1080 and is_ss:
            import glob
            found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            if found:
                rec = json.load(open(found[-1]))
                traffic, traffic_src = rec["hbm_bytes_per_frame"], os.path.relpath(found[-1], ROOT)
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "frame trace = ct_sample_grid + ct_walk_packet<primary> + ct_walk_packet<shadow> + ct_shade "
                              "(the two ct_walk_packet launches are ~95 % of it; rocprof per-kernel split in profiles/)",
                    "kernel_ms": round(kernel_ms, 4),
                    "traffic_rate": (round(traffic / (kernel_ms * 1e-3) / 1e9, 1) if traffic else None),   # measured HBM GB/s
                    "bytes_per_launch": int(launch_bytes), "bytes_per_ray": round(frame_bytes / max(work.rays, 1), 1),
                    "node_visits_per_ray": round(work.node_visits / max(work.rays, 1), 2),
                    "triangle_tests_per_ray": round(work.triangle_tests / max(work.rays, 1), 2)}
        out = {
            "metric": "Mrays/sec + ms/frame, scene_import_dragon.json @1920x1080" if args.scene == "dragon"
                      else "Mrays/sec + ms/frame, %s @%dx%d" % (SCENE_FILES[args.scene], args.width, args.height),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (deterministic 871,200-triangle stand-in for the undistributed "
                                    "Stanford dragon)" if args.scene == "dragon" else "reference asset",
            "config": {"workload": "%s %dx%d T=%d %s, 3 lights with shadow rays" %
                                   (SCENE_FILES[args.scene], args.width, args.height, frame.num_threads,
                                    "supersampling 16 spp (hash jitter)" if is_ss else "1 spp"),
                       "rays_per_frame": int(rays_per_frame), "row_tiles": "8 rows, round-robin over %d GPU(s)" % world,
                       "collective": "RCCL gather to rank 0" if world > 1 else "none"},
            "frames_in_flight": inflight, "ms_frame_latency": round(latency_ms, 4),
            "ms_per_frame_to_host": round(to_host_ms, 4),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            mrays, threads, picks, rays, seconds = cpu_baseline(scene, frame, args.cpu_bands)
            out["cpu_baseline"] = {"value": round(mrays, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                   "sample": "row bands %s of %d of the same frame (%d rays, %.1f s wall), one worker "
                                             "thread per band, oracle/ct_oracle.c" %
                                             (picks, frame.num_threads, rays, seconds),
                                   "host_cores": os.cpu_count()}
        print(json.dumps(out), flush=True)
    for r in renderers:
        r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
