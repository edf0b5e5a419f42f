#!/usr/bin/env python3
"""bench.py -- Mrays/s and ms/frame of the CobbleTrace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is ONE FRAME of BASELINE.json's headline workload: scene_import_dragon.json at 1920x1080 with
the scene file's own settings (32 row bands, supersampling on = 16 jittered samples per pixel,
subsampling off, three lights with shadow rays), jitter from the counter-based generator (the mode in
which GPU and CPU pixels are comparable).  The Stanford dragon itself is not distributed
with the reference; the mesh is the repository's deterministic 871,200-triangle stand-in
(tools/make_dragon_standin.py) -- "data" says so.

What is timed (dispatch -> last pixel in the HOST framebuffer): every frame is traced,
gathered (N > 1), assembled and copied into a pinned host framebuffer; `value` = rays traced by all
ranks / wall time of K frames (max over ranks), scene resident in HBM, a few frames in flight so that
frame i's tail, gather and copy overlap frame i+1's tracing.  The device-only rate is reported beside it.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent --
before anything of torch.cuda or HIP is imported -- runs `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child process, relays the
child's JSON line and exit code, and never touches the GPU (`--dry-run-launch` prints that command instead).  Under
torch.distributed.run (the other way the driver may start it) the ranks are already there.

For N > 1 (one rank per GPU) the frame's row tiles are dealt
round-robin to the ranks and assembled on rank 0 after ONE RCCL gather, issued natively by libctgpu.so
(ctgpu_render_sharded: ncclGather on the frame's stream); the total work per step is fixed, so scaling
is "strong".  torch.distributed only carries the communicator id, the barriers and the timing reduction.
"""
import argparse
import hashlib
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
# the kernels' own records (ct_device.h): QuadNode 128 B, TriRecord 80 B
B_QUAD, B_RECORD = 128, 80
KERNELS = ("ct_sample_grid", "ct_walk_packet<primary>", "ct_light_gate", "ct_walk_packet<shadow>", "ct_shade")
KERNEL_SOURCES = ("ct_pipeline.hip", "ct_device.h", "ct_params.h", "ct_raygen.h", "ct_spheres.h")      # the device code of the fast path


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "bunny", "import", "cube"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--supersampling", type=int, default=-1, help="-1: scene file's setting")
    ap.add_argument("--inflight", type=int, default=0, help="frames in flight (1 = strictly serial frames; default 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--orbit", type=int, default=16,
                    help="views of the moving-camera leg (value_moving_camera: the same K frames, every frame another of this many "
                         "rotations of the camera, the way a viewer re-renders -- raythread.cpp:557-572); 0: no such leg")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="budget of the CPU baseline leg (it drops variants to fit)")
    ap.add_argument("--collective", default="native", choices=["native", "torch", "host"],
                    help="N > 1: the RCCL gather issued by libctgpu.so (default), by torch.distributed, or -- a REHEARSAL for boxes "
                         "with fewer GPUs than ranks, where RCCL cannot serve (it will not put two ranks on one device) -- tiles "
                         "copied to pinned host memory and gathered over gloo")
    ap.add_argument("--same-device", action="store_true",
                    help="every rank uses device 0 (with --collective host: N real ranks, real kernels and partitions on a one-GPU box)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the last timed frame's host framebuffer there (.npy)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: still go through communicator set-up and the gather (a world of one)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the torch.distributed.run command `--gpus N` would start, as one JSON line, and exit")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU, nothing rendered or measured: the ranks only rendezvous over gloo and rank 0 prints a "
                         "record marked as a rehearsal -- checks the launcher, the relay and the per-rank gather on a CPU box")
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_ranks, argv, port=None):
    """The command that runs this script as n_ranks ranks of one node (one per GPU, RCCL over xGMI)."""
    rest = [a for a in argv if a != "--dry-run-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + rest


def launch_ranks(args, argv):
    """`python bench.py --gpus N` typed without a launcher: be the launcher.  This process must not initialise
    the GPU (it would hold a HIP context beside rank 0's, and the box forbids exec after that anyway): nothing
    of torch is imported here, the ranks run in a CHILD process tree, their one JSON line is relayed."""
    import subprocess
    cmd = launch_command(args.gpus, argv)
    if args.dry_run_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    record = None
    for line in proc.stdout:                                   # stderr goes straight through; stdout is filtered
        at = line.find('{"metric"')                            # (other writers' text may share the line)
        if at >= 0:
            try:
                json.loads(line[at:])
                record = line[at:].strip()
                line = line[:at]
            except ValueError:
                pass
        if line.strip():
            sys.stderr.write(line if line.endswith("\n") else line + "\n")
    rc = proc.wait()
    if record is not None:
        print(record, flush=True)
    elif rc == 0:
        print("bench.py: the ranks finished without printing a record", file=sys.stderr)
        rc = 1
    return rc


SCENE_FILES = {"dragon": "scene_import_dragon.json", "bunny": "scene_import_bunny.json",
               "import": "scene_import.json", "cube": "scene_file_cube.json"}


def kernel_source_hash():
    """Identifies the kernels a profiles/ record was measured on (the GPU box has no .git)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "cobbletrace_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profile_record(kind):
    """Newest profiles/rNN_<kind>.json whose kernel_source_hash is that of the sources on disk, else None:
    PMC counters cannot be collected from inside this process, and a stale file must not be quoted."""
    import glob
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % kind)), reverse=True):
        try:
            rec = json.load(open(path))
        except ValueError:
            continue
        if rec.get("kernel_source_hash") == want:
            rec["_source"] = os.path.relpath(path, ROOT)
            return rec
    return None


def cpu_baseline(ca, scene, frame, gpu_host_fb, budget_s):
    """The oracle -- a plain-C port of the reference's Boss/Worker path, one worker thread per row band like
    raythread.cpp:596-618 -- timed on this box's host cores, WHOLE frames of the same workload:
    the scene file's configuration first (its frame is also compared with the GPU's, pixel by pixel), then
    the other thread counts / sampling / jitter modes while the time budget lasts."""
    from oracle import oracle           # the checker, used here only as the timed CPU baseline + parity check
    import copy
    import numpy as np
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    spent = [0.0]

    def run(label, ss, rng, threads):
        f = copy.copy(frame)
        f.flags = (f.flags & ~ca.F_SUPERSAMPLE) | (ca.F_SUPERSAMPLE if ss else 0)
        f.rng_mode = rng
        f.rng_seed = frame.rng_seed if rng == ca.RNG_HASH else 1
        f.num_threads = threads
        t0 = time.perf_counter()
        fb, st = oracle.render(scene.desc, f, os_threads=0)          # one OS thread per band, as the reference
        s = time.perf_counter() - t0
        spent[0] += s
        return {"config": label, "threads": threads, "rays": int(st.rays), "seconds": round(s, 3),
                "Mrays/s": round(st.rays / s / 1e6, 3), "ms/frame": round(s * 1e3, 1)}, fb

    T = frame.num_threads
    is_ss = bool(frame.flags & ca.F_SUPERSAMPLE)
    head, fb = run("scene file: ss=%s, T=%d, hash jitter" % ("on" if is_ss else "off", T), is_ss, ca.RNG_HASH, T)
    parity = None
    if gpu_host_fb is not None:
        a, b = gpu_host_fb, fb
        shifts = np.array([0, 8, 16])
        diff = np.abs(((a[..., None] >> shifts) & 255).astype(np.int32) - ((b[..., None] >> shifts) & 255).astype(np.int32))
        parity = {"compared": "whole frame, GPU host framebuffer of the last timed frame vs oracle (%d x %d)" % a.shape[::-1],
                  "differing_pixels": int((a != b).sum()), "parity_max_lsb": int(diff.max()),
                  "pixels_written": int((b != 0).sum())}
    variants = [head]
    plan = [("ss=%s, T=%d, hash jitter" % ("off" if is_ss else "on", T), not is_ss, ca.RNG_HASH, T),
            ("ss=on, T=nproc=%d, hash jitter" % min(cores, frame.height), True, ca.RNG_HASH, min(cores, frame.height)),
            ("ss=off, T=nproc=%d" % min(cores, frame.height), False, ca.RNG_HASH, min(cores, frame.height)),
            ("ss=on, T=1, libc rand() (the reference's own generator; deterministic only at T=1)", True, ca.RNG_LIBC, 1)]
    for label, ss, rng, threads in plan:
        # a variant costs about the head's time scaled by its ray count and thread count
        guess = head["seconds"] * (1.0 if ss == is_ss else (1 / 16.0 if is_ss else 16.0)) * (max(1, min(T, cores)) / max(1, min(threads, cores)))
        if spent[0] + guess > budget_s:
            variants.append({"config": label, "skipped": "over the --cpu-seconds budget (estimated %.0f s)" % guess})
            continue
        variants.append(run(label, ss, rng, threads)[0])
    out = {"value": head["Mrays/s"], "unit": "Mrays/s", "cores": min(T, cores), "kind": "port",
           "sample": "the WHOLE frame of the same workload (%d rays in %.2f s): oracle/ct_oracle.c, one worker thread per row "
                     "band (T = %d threads on the %d host cores this process may use)" % (head["rays"], head["seconds"], T, cores),
           "ms_per_frame": head["ms/frame"], "host_cores": cores, "variants": variants,
           "note": "libc mode draws rand() into a table up front, so it does not reproduce the reference's rand() lock "
                   "contention (BASELINE.md: 32 reference threads are slower than 1 with supersampling on)"}
    return out, parity


def rehearse(args):
    """The plumbing of an N-rank run without a GPU: rendezvous (gloo), one all_gather of a per-rank row, one
    record from rank 0 on stdout, noise on stdout from the other ranks that the relay must drop."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    mine = torch.tensor([float(rank), 100.0 + rank], dtype=torch.float64)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
    else:
        rows = [mine]
    print("rank %d: this line is not the record" % rank, flush=True)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launch rehearsal (nothing measured)", "value": None, "n_gpus": world,
                          "ranks": [[int(r[0]), r[1].item()] for r in rows]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if (args.gpus > 1 and "WORLD_SIZE" not in os.environ) or args.dry_run_launch:
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args, sys.argv[1:]))
        print(json.dumps({"launch": None, "note": "--gpus 1 runs in this process"}))
        return
    if args.rehearse_launch:
        return rehearse(args)
    # stdout carries exactly ONE line, the JSON record: everything else that may write to file descriptor 1 on the
    # way (RCCL prints a version banner there when a communicator comes up) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    import cobbletrace_amd as ca
    from cobbletrace_amd import multi_gpu
    import __graft_entry__ as entry

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # (CT_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank rehearses every collective of the N-rank
    # path on a one-GPU box; RCCL will not put two ranks on one device)
    use_dist = world > 1 or (bool(os.environ.get("CT_BENCH_FORCE_DIST")) and "RANK" in os.environ)
    host_collective = args.collective == "host"
    if args.same_device and world > 1 and not host_collective:
        raise SystemExit("--same-device with several ranks needs --collective host (RCCL does not put two ranks on one device)")
    # tensors torch.distributed reduces: on the device under RCCL, on the host under gloo
    red_device = torch.device("cpu") if host_collective else device
    if use_dist:
        if host_collective:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    # every rank needs the assets and the native libraries; rank 0 prepares, the others wait
    if rank == 0:
        libs = [os.path.join(ROOT, "cobbletrace_amd", "csrc", "libctgpu.so"),
                os.path.join(ROOT, "cobbletrace_amd", "host", "libcthost.so"),
                os.path.join(ROOT, "oracle", "libctoracle.so")]
        if not all(os.path.exists(p) for p in libs):
            entry.build()
        entry.prepare_assets(need_dragon=(args.scene == "dragon"))
    if use_dist:
        dist.barrier()
    entry.check_build_ids()           # the libraries that are about to be measured were made from the sources on disk

    scene = ca.HostScene(os.path.join(ROOT, "assets", SCENE_FILES[args.scene]))
    ss = None if args.supersampling < 0 else bool(args.supersampling)
    frame = scene.frame(args.width, args.height, supersampling=ss, rng_mode=ca.RNG_HASH, rng_seed=0)
    frame_still = frame
    is_ss = bool(frame.flags & ca.F_SUPERSAMPLE)
    tile_rows = 8
    lights = [scene.desc.lights[i] for i in range(scene.desc.n_lights)]
    n_shadow_lights = sum(1 for l in lights if l.type in (ca.LIGHT_POINT, ca.LIGHT_DIRECTIONAL))       # one shadow ray per hit each

    # A few frames in flight: frame i+1 is traced on another stream / context while frame i's kernels drain
    # their tail and its tiles are gathered, assembled and copied to the host, the way a renderer producing a
    # stream of frames runs.  Each slot owns a context (scene copy + scratch arrays), a stream, a communicator,
    # a device framebuffer and a pinned host framebuffer.
    # (measured on one GPU with rank 0's share of the frame, a round-1 measurement, repeated by tools/rank_share.py: an eighth of the frame takes 0.49 ms with
    # three frames in flight and 0.46 ms with four; the whole frame was best with three until round 4's shorter shadow walk:
    # profiles/r04_inflight_sweep.log, one GPU, whole frame: 2 / 3 / 4 / 5 in flight = 3.05 / 2.66 / 2.62 / 2.71 ms per frame)
    inflight = args.inflight if args.inflight > 0 else 4
    # Throughput tuning through the library's documented knobs (read at context creation): with several
    # frames in flight and only an eighth of the frame per rank, a smaller persistent grid per context
    # and 128-ray-id chunks beat the defaults, which are tuned for one frame's latency (DESIGN.md section 6;
    # profiles/r05_rank_share.log, rank 0's share rehearsed on one GPU: a quarter of the frame 0.62 -> 0.56 ms, an eighth
    # 0.47 -> 0.30 ms, a half and the whole frame the same either way).
    if inflight > 1 and world >= 4:
        os.environ.setdefault("CTGPU_BLOCKS_PER_CU", "12")
        os.environ.setdefault("CTGPU_CHUNK", "128")

    use_comm = use_dist or args.force_collective
    collective = "none"
    native_ok = True

    def all_ranks_agree(flag):
        t = torch.tensor([flag], device=red_device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.item() >= 1.0

    slots = []
    for i in range(inflight):
        ctx = ca.GpuContext(local_rank)
        if i == 0:          # vertices in, BuildBVH + the walks' layouts made on the device; ONE copy per rank, shared by the slots
            t_up = time.perf_counter()
            ctx.upload(scene.desc, build_on_device=True)
            scene_setup_ms = (time.perf_counter() - t_up) * 1e3
            scene_info = ctx.scene_info()
        else:
            ctx.share_scene(slots[0]["ctx"])
        part = ca.Partition(tile_rows, rank, world)
        n_tile = ctx.partition_pixels(frame, part)
        slot = {"ctx": ctx, "part": part, "stream": torch.cuda.Stream(device=device), "n_tile": n_tile,
                "tiles": torch.zeros(n_tile, dtype=torch.int32, device=device),
                # --collective host: this rank's tiles on the host (pinned), everybody's on rank 0's host and device
                "host_tiles": torch.zeros(n_tile, dtype=torch.int32).pin_memory() if host_collective else None,
                "host_all": torch.zeros(n_tile * world, dtype=torch.int32).pin_memory() if host_collective and rank == 0 else None,
                "dev_all": torch.zeros(n_tile * world, dtype=torch.int32, device=device) if host_collective and rank == 0 else None,
                "fb": torch.zeros(frame.height * frame.width, dtype=torch.int32, device=device) if rank == 0 else None,
                "host_fb": torch.zeros(frame.height * frame.width, dtype=torch.int32).pin_memory() if rank == 0 else None,
                "native": False}
        if use_comm and args.collective == "native" and native_ok and not host_collective:
            # rank 0 makes the RCCL unique id, torch.distributed carries its 128 bytes to the other ranks (one
            # communicator per rank, shared by the slots: RCCL then orders the in-flight frames' gathers itself).
            # Every step that can fail on one rank only is followed by a MIN over the ranks, so that all ranks take
            # the same branch and nobody waits in a broadcast or an init the others have left.
            if i == 0:
                uid = torch.zeros(128, dtype=torch.uint8, device=device)
                have_id = 1.0
                if rank == 0:
                    try:
                        uid.copy_(torch.frombuffer(bytearray(ca.GpuContext.comm_unique_id()), dtype=torch.uint8))
                    except RuntimeError as e:
                        have_id = 0.0
                        print("bench.py: native communicator unavailable (%s); gathering through torch.distributed" % e, file=sys.stderr)
                native_ok = all_ranks_agree(have_id)
                if native_ok:
                    if use_dist:
                        dist.broadcast(uid, src=0)
                    try:
                        ctx.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
                        up = 1.0
                    except RuntimeError as e:
                        up = 0.0
                        print("bench.py: rank %d: ctgpu_comm_init failed (%s)" % (rank, e), file=sys.stderr)
                    native_ok = all_ranks_agree(up)
            else:
                ctx.comm_share(slots[0]["ctx"])
            slot["native"] = native_ok
        slots.append(slot)
    if use_comm:
        if not native_ok:
            for s in slots:
                s["native"] = False
        collective = ("RCCL ncclGather to rank 0, issued by libctgpu.so (ctgpu_render_sharded)" if slots[0]["native"]
                      else "REHEARSAL: tiles through pinned host memory, gathered to rank 0 over gloo" +
                           (" (all ranks on device 0)" if args.same_device else "") if host_collective
                      else "RCCL gather to rank 0 through torch.distributed")

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    phase_events = []          # timed region only: (kind, start, end) torch events on the slot's stream

    def mark(stream):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    # The moving-camera leg's views: the camera turned about its vertical and horizontal axes in steps of pi/64 (a quarter
    # of the reference's key step, raythread.cpp:405-420: a viewer re-renders only when the camera moved, :557-562), a
    # fixed list every rank walks in the same order.  The headline renders ONE view over and over, which keeps the same
    # nodes in the caches; this says what a moving viewer gets.
    views = []
    for k in range(max(args.orbit, 0)):
        import copy as _copy
        f = _copy.copy(frame)
        yaw = ((k * 5) % args.orbit - args.orbit // 2) * (np.pi / 64) * (8.0 / max(args.orbit, 8))
        pitch = ((k * 3) % 5 - 2) * (np.pi / 128)
        f.cam_rot = ca.camera_rotation(yaw, pitch, 0.0)
        views.append(f)

    def one_frame(i, to_host=True, record=False, moving=False):
        """Issue frame i: trace my tiles, gather to rank 0, assemble there, copy to the host framebuffer -- all
        asynchronous on the slot's stream."""
        s = slots[i % inflight]
        stream = s["stream"]
        frame = views[i % len(views)] if moving else frame_still
        with torch.cuda.stream(stream):
            if s["native"]:         # (its phases are timed inside the library: ctgpu_read_phase_times)
                s["ctx"].render_sharded(frame, tile_rows, s["fb"].data_ptr() if rank == 0 else 0, stream.cuda_stream)
            else:
                e0 = mark(stream) if record else None
                s["ctx"].render_partition(frame, s["part"], s["tiles"].data_ptr(), stream.cuda_stream)
                e1 = mark(stream) if record else None
                if host_collective and use_dist:
                    # the rehearsal's hand-over: device -> pinned host, gloo gather, rank 0: host -> device (the frame's
                    # stream waits for its own copy; the other frames in flight keep tracing meanwhile)
                    s["host_tiles"].copy_(s["tiles"], non_blocking=True)
                    stream.synchronize()
                    pieces = list(s["host_all"].split(s["n_tile"])) if rank == 0 else None
                    dist.gather(s["host_tiles"], gather_list=pieces, dst=0)
                    gathered = None
                    if rank == 0:
                        s["dev_all"].copy_(s["host_all"], non_blocking=True)
                        gathered = s["dev_all"]
                else:
                    gathered = s["tiles"] if not use_dist else multi_gpu.gather_to_root(s["tiles"], world, rank)
                e2 = mark(stream) if record else None
                if rank == 0:
                    s["ctx"].assemble(frame, s["part"], gathered.data_ptr(), s["fb"].data_ptr(), stream.cuda_stream)
                if record:
                    phase_events.extend([("trace", e0, e1), ("gather", e1, e2), ("assemble", e2, mark(stream))])
            if rank == 0 and to_host:       # the pixels the reference writes, into the pinned host framebuffer
                e3 = mark(stream) if record else None
                s["ctx"].download_frame(frame, s["fb"].data_ptr(), s["host_fb"].data_ptr(), stream.cuda_stream)
                if record:
                    phase_events.append(("d2h", e3, mark(stream)))
        return s

    def timed(n_frames, to_host, record=False, moving=False):
        sync_all()
        t0 = time.perf_counter()
        for i in range(n_frames):
            one_frame(i, to_host, record, moving)
        sync_all()
        return time.perf_counter() - t0

    # what the communicator itself reports (ncclCommUserRank / ncclCommCount through ctgpu_comm_rank): [rank, ranks]
    comm_seen = None
    if use_comm and slots[0]["native"]:
        try:
            comm_seen = list(slots[0]["ctx"].comm_rank())
        except RuntimeError as e:
            comm_seen = "ctgpu_comm_rank failed: %s" % e
    # Warm-up.  Its first frames are also the native gather's first real use with this many ranks: should the library's
    # own RCCL path fail on some rank (an error, not a hang), every rank hears of it in the MIN below and all of them go
    # on with the gather through torch.distributed -- a number measured with the second-best collective, and saying so,
    # beats none.
    native_failed = 0.0
    try:
        for i in range(inflight):
            one_frame(i)
        torch.cuda.synchronize()
    except RuntimeError as e:
        if not (use_comm and slots[0]["native"]):
            raise
        native_failed = 1.0
        print("bench.py: rank %d: the native gather failed in the warm-up (%s); falling back to torch.distributed" % (rank, e), file=sys.stderr)
    if use_comm and slots[0]["native"] and not all_ranks_agree(1.0 - native_failed):
        for s in slots:
            s["native"] = False
        collective = "RCCL gather to rank 0 through torch.distributed (the native gather failed in the warm-up)"
    for i in range(max(args.warmup, inflight)):
        one_frame(i)
    sync_all()
    for s in slots:
        s["ctx"].read_counters()             # reset the ray counters
        s["ctx"].read_walk_work()
        s["ctx"].kernel_timing(True)         # HIP events around every pipeline kernel, on the frame's own stream

    # ---- timed region: exactly K frames, dispatch -> pixels in the host framebuffer
    elapsed = timed(args.steps, to_host=True, record=True)

    counted = [s["ctx"].read_counters() for s in slots]
    times = [s["ctx"].read_kernel_times() for s in slots]
    # (the last TIMED frame, as it arrived in the pinned host framebuffer: what the parity check compares -- the legs below draw over it)
    headline_last = slots[(args.steps - 1) % inflight]["host_fb"].numpy().view(np.uint32).reshape(frame.height, frame.width).copy() if rank == 0 else None
    phases = {"trace": 0.0, "gather": 0.0, "assemble": 0.0, "d2h": 0.0}
    for s in slots:
        if s["native"]:
            pt = s["ctx"].read_phase_times()
            phases["trace"] += pt.trace_ms; phases["gather"] += pt.gather_ms; phases["assemble"] += pt.assemble_ms
    for kind, a, b in phase_events:
        phases[kind] += a.elapsed_time(b)
    works = [s["ctx"].read_walk_work() for s in slots]
    for s in slots:
        s["ctx"].kernel_timing(False)
    my_rays = float(sum(c.rays for c in counted))
    my_elided = float(sum(c.rays_shadow_elided for c in counted))
    totals = torch.tensor([my_rays, my_elided, elapsed], dtype=torch.float64, device=red_device)
    if use_dist:
        sums = totals[:2].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        time_max = totals[2:].clone()
        dist.all_reduce(time_max, op=dist.ReduceOp.MAX)
        total_rays, total_elided, elapsed = float(sums[0].item()), float(sums[1].item()), float(time_max.item())
    else:
        total_rays, total_elided = my_rays, my_elided
    # per-rank breakdown (means per frame): what a first scaling curve is read with
    mine = torch.tensor([my_rays / args.steps, phases["trace"] / args.steps, phases["gather"] / args.steps,
                         phases["assemble"] / args.steps, phases["d2h"] / args.steps, float(slots[0]["n_tile"]),
                         totals[2].item() / args.steps * 1e3], dtype=torch.float64, device=red_device)
    if use_dist:
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = [r.cpu().tolist() for r in rows]
    else:
        per_rank = [mine.cpu().tolist()]
    rays_per_frame = total_rays / args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / elapsed / 1e6

    # ---- outside the timed region: the same K frames left on the device (no host copy), and one frame alone
    elapsed_device = timed(args.steps, to_host=False)
    if use_dist:
        t = torch.tensor([elapsed_device], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_device = float(t.item())
    # ---- the moving-camera leg: the same K frames, delivered to the host like the headline's, every frame another view
    moving = None
    if views:
        for i in range(len(views)):              # (every view once, untimed: first-touch effects are not what is asked)
            one_frame(i, True, False, True)
        sync_all()
        for s in slots:
            s["ctx"].read_counters()
        elapsed_moving = timed(args.steps, to_host=True, moving=True)
        moved = [s["ctx"].read_counters() for s in slots]
        t = torch.tensor([float(sum(c.rays for c in moved)), elapsed_moving], dtype=torch.float64, device=red_device)
        if use_dist:
            rays_sum = t[:1].clone(); dist.all_reduce(rays_sum, op=dist.ReduceOp.SUM)
            t_max = t[1:].clone(); dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
            t = torch.cat([rays_sum, t_max])
        moving = {"rays": float(t[0].item()), "elapsed": float(t[1].item()), "last_view": (args.steps - 1) % len(views)}
        if rank == 0:
            moving["last_frame"] = slots[(args.steps - 1) % inflight]["host_fb"].numpy().view(np.uint32).reshape(frame.height, frame.width).copy()
    serial = 5
    slots[0]["ctx"].kernel_timing(True)
    sync_all()
    t0 = time.perf_counter()
    for k in range(serial):
        one_frame(0)
        sync_all()
    latency_ms = (time.perf_counter() - t0) / serial * 1e3
    alone = slots[0]["ctx"].read_kernel_times()          # the same kernels with nothing else on the GPU
    slots[0]["ctx"].kernel_timing(False)
    for s in slots:
        s["ctx"].read_counters()

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel, from what was measured live in the timed region
        launches = [sum(t.launches[k] for t in times) for k in range(5)]
        avg_ms = [sum(t.total_ms[k] for t in times) / max(launches[k], 1) for k in range(5)]
        n_launch = max(launches[1], 1)
        packets = [sum(w.packets[k] for w in works) / n_launch for k in range(2)]
        quads = [sum(w.quad_nodes[k] for w in works) / n_launch for k in range(2)]
        records = [sum(w.triangle_records[k] for w in works) / n_launch for k in range(2)]
        per_launch_primary = sum(c.rays_primary for c in counted) / n_launch
        per_launch_shadow = sum(c.rays_shadow for c in counted) / n_launch
        per_launch_elided = sum(c.rays_shadow_elided for c in counted) / n_launch
        n_lights = n_shadow_lights if per_launch_shadow else 0
        samples = per_launch_primary                                       # sample slots of this rank's share (traced ones)
        pixels = samples / (16 if is_ss else 1)
        hits = (per_launch_shadow + per_launch_elided) / max(n_lights, 1)
        # bytes each kernel itself asks the memory system for, per launch (ct_device.h record sizes; per-sample
        # arrays: position 8 B, hit record 8 B, one shadow byte per light)
        own = {
            "ct_sample_grid": samples * 8 if is_ss else 0,
            "ct_walk_packet<primary>": quads[0] * B_QUAD + records[0] * B_RECORD + samples * ((8 if is_ss else 0) + 8),
            "ct_light_gate": samples * (8 + (8 if is_ss else 0) + n_lights) + hits * 60,
            "ct_walk_packet<shadow>": quads[1] * B_QUAD + records[1] * B_RECORD + samples * n_lights * 1
                                      + per_launch_shadow * ((8 if is_ss else 0) + 8 + 1),
            "ct_shade": samples * (8 + (8 if is_ss else 0) + n_lights) + hits * 60 + pixels * 4,
        }
        alone_ms = [alone.total_ms[k] / max(alone.launches[k], 1) for k in range(5)]
        # (the shadow rays' kernel by the name rocprofv3 shows it under: ct_walk_shadow_tree where the shadow tree took them)
        path = counted[0].path()
        KERNELS = list(globals()["KERNELS"])
        if path["shadow_tree"]:
            KERNELS[3] = "ct_walk_shadow_tree"
            own["ct_walk_shadow_tree"] = own.pop("ct_walk_packet<shadow>")
        if path["closest_tree"]:
            KERNELS[1] = "ct_walk_closest_tree"
            own["ct_walk_closest_tree"] = own.pop("ct_walk_packet<primary>")
        kernels = {KERNELS[k]: {"avg_ms": round(alone_ms[k], 4), "avg_ms_inflight": round(avg_ms[k], 4), "launches_inflight": int(launches[k]),
                                "own_bytes": int(own[KERNELS[k]]),
                                "own_GB/s": round(own[KERNELS[k]] / max(alone_ms[k], 1e-9) / 1e6, 1)} for k in range(5)}
        # the dominant kernel: by its UNDISTURBED duration (serial frames), the figure profiles/rNN_kernel_stats.csv
        # (rocprofv3 of `bench.py --inflight 1`) reproduces; the duration inside the timed region, where kernels of the
        # frames in flight share the GPU, is reported beside it
        dom = max(range(5), key=lambda k: alone_ms[k])
        dom_name = KERNELS[dom]
        achieved = own[dom_name] / (alone_ms[dom] * 1e-3) / 1e9
        achieved_inflight = own[dom_name] / (avg_ms[dom] * 1e-3) / 1e9
        traffic_rec, issue_rec = profile_record("traffic"), profile_record("issue")
        traffic = None
        headline = world == 1 and args.scene == "dragon" and (args.width, args.height) == (1920, 1080) and is_ss
        if traffic_rec and headline:
            traffic = traffic_rec.get("hbm_bytes_per_launch", {}).get(dom_name)
        # What actually limits the walk: the vector pipe's issue slots -- counters of a profiled
        # run of the same kernels (tools/collect_profiles.sh -> profiles/rNN_issue.json, quoted only when the kernel
        # sources are the ones on disk); "hbm" above is the bound the contract names for this path, not the limiter.
        limiter = {"what": "valu-issue", "valu_pipe_busy_frac": None, "valu_lane_utilisation": None, "wave_cycles_in_waitcnt_frac": None,
                   "source": None, "note": "the walks are bound by vector-instruction issue and each wave's dependent chain "
                                           "(fetch -> four box tests -> decide), not by bytes: DESIGN.md section 4"}
        if issue_rec and headline and dom_name in issue_rec.get("kernels", {}):
            rec = issue_rec["kernels"][dom_name]
            limiter.update({k: rec.get(k) for k in ("valu_pipe_busy_frac", "valu_lane_utilisation", "wave_cycles_in_waitcnt_frac")})
            limiter["source"] = issue_rec["_source"]
        # the reference-layout figure of SURVEY.md 8(d), whole frame, instrumented kernel (not the bound: see DESIGN.md);
        # per walk: the same scene without its lights traces the primary rays alone
        work = work_primary = None
        if world == 1:
            work = slots[0]["ctx"].count_work(frame)
            import ctypes
            bare = ca.SceneDesc()
            ctypes.memmove(ctypes.byref(bare), ctypes.byref(scene.desc), ctypes.sizeof(bare))      # (same arrays, no lights)
            bare.n_lights = 0
            ctx_bare = ca.GpuContext(local_rank)
            try:
                ctx_bare.upload(bare, build_on_device=True)
                work_primary = ctx_bare.count_work(frame)
            finally:
                ctx_bare.close()
        roofline = {"bound": "hbm", "limiter": limiter, "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": (traffic_rec["_source"] if traffic is not None else None),
                    "launch_ms": round(alone_ms[dom], 4), "launches_timed": int(alone.launches[dom]),
                    "timing": "HIP events around each kernel on the frame's stream; launch_ms / achieved / frac: the kernel with the "
                              "GPU to itself (%d serial frames right after the timed region) -- what rocprofv3 --kernel-trace of "
                              "`bench.py --inflight 1` shows (profiles/); *_inflight: the same launches inside the timed region, "
                              "%d frames in flight, where kernels of different frames share the GPU and each lasts longer" % (serial, inflight),
                    "launch_ms_inflight": round(avg_ms[dom], 4), "launches_timed_inflight": int(launches[dom]),
                    "achieved_inflight": round(achieved_inflight, 1), "frac_inflight": round(achieved_inflight / HBM_PEAK_GBS, 4),
                    "bytes_per_launch": int(own[dom_name]),
                    "bytes_model": "the kernel's own requests: QuadNodes fetched x 128 B + triangle records fetched x 80 B "
                                   "(counted by the kernel) + its per-sample / per-ray arrays" +
                                   (" (%s: the shadow tree's nodes and its own copy of the records, same sizes)" % dom_name if dom_name in ("ct_walk_shadow_tree", "ct_walk_closest_tree") else ""),
                    "per_launch": {"rays": int(per_launch_shadow if dom == 3 else per_launch_primary),
                                   "packets": int(packets[1 if dom == 3 else 0]), "quad_nodes": int(quads[1 if dom == 3 else 0]),
                                   "triangle_records": int(records[1 if dom == 3 else 0]),
                                   "shadow_rays_blocked": int(sum(w.shadow_rays_blocked for w in works) / n_launch),
                                   "shadow_packets_all_blocked": int(sum(w.shadow_packets_all_blocked for w in works) / n_launch)},
                    "kernels": kernels}
        if work is not None:
            def layout_bytes(visits, tests, hits_, pixels_):
                return visits * B_NODE + tests * B_TRI + hits_ * B_HIT + pixels_ * B_PIXEL
            n_pix = int(work.rays_primary // (16 if is_ss else 1))
            ref_bytes = layout_bytes(work.node_visits, work.triangle_tests, work.rays_shadow // max(n_shadow_lights, 1), n_pix)
            frame_kernel_ms = sum(alone_ms)
            roofline["reference_bytes"] = {
                "per_frame": int(ref_bytes), "per_ray": round(ref_bytes / max(work.rays, 1), 1),
                "node_visits_per_ray": round(work.node_visits / max(work.rays, 1), 2),
                "triangle_tests_per_ray": round(work.triangle_tests / max(work.rays, 1), 2),
                "rate_GB/s": round(ref_bytes / (frame_kernel_ms * 1e-3) / 1e9, 1),
                "note": "SURVEY.md 8(d): per-ray node / triangle fetches of the reference layout over the frame's kernel time "
                        "(undisturbed launches); above the HBM peak because a wave fetches a node once for its 64 rays -- "
                        "bytes the design does not move"}
            if work_primary is not None:
                vis_p, tst_p = work_primary.node_visits, work_primary.triangle_tests
                vis_s, tst_s = work.node_visits - vis_p, work.triangle_tests - tst_p
                for label, k, vis, tst, n_rays in (("primary_walk", 1, vis_p, tst_p, work.rays_primary), ("shadow_walk", 3, vis_s, tst_s, work.rays_shadow)):
                    b = vis * B_NODE + tst * B_TRI
                    roofline["reference_bytes"][label] = {
                        "bytes": int(b), "node_visits": int(vis), "triangle_tests": int(tst), "rays": int(n_rays),
                        "per_ray": round(b / max(n_rays, 1), 1), "launch_ms": round(alone_ms[k], 4),
                        "rate_GB/s": round(b / max(alone_ms[k], 1e-9) / 1e6, 1)}
        if issue_rec:
            roofline["issue"] = {k: v for k, v in issue_rec.items() if not k.startswith("_") and k != "kernel_source_hash"}
            roofline["issue"]["source"] = issue_rec["_source"]
        out = {
            "metric": "Mrays/sec + ms/frame, scene_import_dragon.json @1920x1080" if args.scene == "dragon"
                      else "Mrays/sec + ms/frame, %s @%dx%d" % (SCENE_FILES[args.scene], args.width, args.height),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (deterministic 871,200-triangle stand-in for the undistributed "
                                    "Stanford dragon)" if args.scene == "dragon" else "reference asset",
            "config": {"workload": "%s %dx%d T=%d %s, %d lights (%d with shadow rays)" %
                                   (SCENE_FILES[args.scene], args.width, args.height, frame.num_threads,
                                    "supersampling 16 spp (hash jitter)" if is_ss else "1 spp", len(lights), n_shadow_lights),
                       "rays_per_frame": int(rays_per_frame),
                       "reference_rays_per_frame": int(rays_per_frame + total_elided / args.steps),
                       "rays_note": "value counts rays actually traced; the reference also starts the shadow rays "
                                    "ct_light_gate proves irrelevant (reference_rays_per_frame)",
                       "row_tiles": "%d rows, round-robin over %d GPU(s)" % (tile_rows, world), "collective": collective},
            "timed": "dispatch -> pixels in the pinned host framebuffer, every frame",
            "build_id": {"libctgpu": ca.gpu_library().ctgpu_build_id().decode(), "libcthost": ca.host_library().cth_build_id().decode(),
                         "kernel_source_hash": kernel_source_hash()},
            "frames_in_flight": inflight, "ms_frame_latency": round(latency_ms, 4),
            "path": dict(path, note="which code rendered the timed frames (ctgpu_stats): several paths give the same pixels, the context "
                                    "picks by scene and frame kind; shadow_rays_left: per read, rays the shadow tree left to the reference-tree launch"),
            "shadow_tree": {"nodes": int(scene_info.shadow_tree_nodes), "stack": int(scene_info.shadow_tree_pending),
                            "build_ms": round(scene_info.shadow_tree_ms, 2), "gamma": scene_info.shadow_tree_gamma},
            "scene_setup": {"what": "vertices in host memory -> BuildBVH + walk layouts in HBM (ctgpu_upload_scene without a tree), "
                                    "once per rank; the frames in flight share that one copy",
                            "wall_ms": round(scene_setup_ms, 2), "device_build_ms": round(scene_info.build_ms, 3),
                            "device_pack_ms": round(scene_info.pack_ms, 3), "hbm_MB": round(scene_info.hbm_bytes / 1e6, 1)},
            "value_device_only": round(total_rays / elapsed_device / 1e6, 2),
            "ms_per_step_device_only": round(elapsed_device / args.steps * 1e3, 4),
            "roofline": roofline,
            "scaling_detail": {
                "note": "means per frame over the timed region, per rank; the phases are HIP events on the frame's stream "
                        "(with several frames in flight they overlap other frames' work, so they do not add up to ms_per_step); "
                        "gather_ms on rank 0 ends when the slowest rank's tiles have arrived; wall_ms = that rank's wall time per frame",
                "tile_rows": tile_rows,
                "collective_used": collective, "native_gather": bool(slots[0]["native"]),
                "rccl_ranks": comm_seen,
                "ranks": [{"rank": r, "rays": int(v[0]), "trace_ms": round(v[1], 4), "gather_ms": round(v[2], 4),
                           "assemble_ms": round(v[3], 4), "d2h_ms": round(v[4], 4), "tile_pixels": int(v[5]),
                           "row_tiles": int(v[5]) // max(1, tile_rows * 2 * (frame.height // 2)), "wall_ms": round(v[6], 4)}
                          for r, v in enumerate(per_rank)]},
        }
        if moving is not None:
            out["value_moving_camera"] = round(moving["rays"] / moving["elapsed"] / 1e6, 2)
            out["ms_per_step_moving_camera"] = round(moving["elapsed"] / args.steps * 1e3, 4)
            out["moving_camera"] = {"views": len(views), "rays_per_frame": int(moving["rays"] / args.steps),
                                    "what": "the timed region once more, every frame another of `views` camera rotations (yaw / pitch in steps "
                                            "of pi/64 about the scene file's view), each frame delivered to the pinned host framebuffer",
                                    "vs_still": round(out["value_moving_camera"] / max(out["value"], 1e-9), 4)}
        if world == 1 and not args.no_cpu_baseline:
            if moving is not None:
                # the last timed frame of the moving-camera leg against the oracle's frame of that view, whole
                from oracle import oracle as _oracle
                want, _ = _oracle.render(scene.desc, views[moving["last_view"]], os_threads=0)
                out["moving_camera"]["parity"] = {"view": moving["last_view"], "differing_pixels": int((want != moving["last_frame"]).sum()),
                                                  "pixels": int(want.size)}
            out["cpu_baseline"], out["parity"] = cpu_baseline(ca, scene, frame, headline_last, args.cpu_seconds)
            out["speedup_vs_cpu_ms_per_frame"] = round(out["cpu_baseline"]["ms_per_frame"] / ms_per_step, 1)
        if args.dump_frame:
            np.save(args.dump_frame, headline_last)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    for s in slots:
        s["ctx"].close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
