"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
inputs.  Bar: <= 1 LSB per 8-bit channel (BASELINE.json); index / t of individual hits exact
(the only non-exact operation on the path is pow(), SURVEY.md 7 "hard parts")."""
import ctypes as C
import os

import numpy as np
import pytest

import cobbletrace_amd as ca
from conftest import channel_diff

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle
    return oracle


def _compare(gpu_fb, cpu_fb, what):
    diff = channel_diff(gpu_fb, cpu_fb)
    n_diff = int((gpu_fb != cpu_fb).sum())
    print("%s: max channel diff %d, pixels differing %d of %d" % (what, diff.max(), n_diff, gpu_fb.size))
    assert diff.max() <= 1, "%s: %d LSB off" % (what, diff.max())      # tolerance: 1 LSB / channel
    return n_diff


class _knob:
    """`with _knob("CTGPU_PACKET", "0"):` -- a CTGPU_* variable for the duration, then whatever was there before
    (the suite itself may be running under a knob: CTGPU_PACKET=0 python -m pytest ...)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.before = os.environ.get(self.name)
        os.environ[self.name] = self.value

    def __exit__(self, *exc):
        if self.before is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.before


def _context_with(monkeypatch, scene, **knobs):
    """A context of its own whose knobs are what THIS test says (CTGPU_* are read when a context is made), whatever
    the environment the suite runs under: None removes a variable."""
    for key, value in knobs.items():
        if value is None:
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, str(value))
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    return ctx


def _random_rays(scene, n, seed):
    rng = np.random.default_rng(seed)
    v = scene.vertices()
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    centre, span = (lo + hi) / 2, (hi - lo).max()
    rays = np.zeros(n, dtype=ca.bindings.RAY_DTYPE)
    origin = centre + rng.normal(size=(n, 3)) * span * 1.5
    target = lo + rng.random((n, 3)) * (hi - lo)
    rays["origin"] = origin
    rays["direction"] = target - origin
    rays["t"] = np.float32(1e30)
    # a few degenerate ones: axis-parallel directions (zero components -> inf/NaN slabs), t = 0 starts
    rays["direction"][: n // 50, 0] = 0.0
    rays["direction"][n // 50: n // 25, 1] = 0.0
    rays["t"][n // 25: n // 20] = 0.0
    return rays


@pytest.mark.parametrize("name", ["cube", "import", "bunny"])
def test_closest_hit_exact(name, scenes, gpu_ctx):
    scene, ctx = scenes(name), gpu_ctx(name)
    rays = _random_rays(scene, 200_000 if name != "cube" else 50_000, seed=11)
    got = ctx.trace_rays(rays)
    want = _oracle().trace_rays(scene.desc, rays)
    assert (got["found"] == want["found"]).all()
    assert (got["triangle"] == want["triangle"]).all()          # exact index, ties included
    assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()          # bit-exact t
    assert (got["tclosest"].view(np.uint32) == want["tclosest"].view(np.uint32)).all()
    assert want["found"].sum() > len(rays) // 20                 # the test really hits geometry


@pytest.mark.parametrize("name,w,h,t", [("cube", 640, 480, 8), ("import", 1920, 1080, 32), ("bunny", 1920, 1080, 32),
                                        ("bunny", 1920, 1080, 8)])
def test_frame_single_sample(name, w, h, t, scenes, gpu_ctx):
    scene, ctx = scenes(name), gpu_ctx(name)
    frame = scene.frame(w, h, threads=t, supersampling=False)
    gpu_fb, st = ctx.render(frame)
    cpu_fb, cst = _oracle().render(scene.desc, frame, os_threads=8)
    _compare(gpu_fb, cpu_fb, "%s %dx%d T=%d" % (name, w, h, t))
    # untouched pixels stay untouched (row 0, rows beyond T*(H/T), columns outside the square)
    assert ((gpu_fb == 0) == (cpu_fb == 0)).all()
    # the GPU does not trace the y = -H/2 row, whose pixels the reference computes and drops
    assert st.rays_primary == cst.rays_primary - 2 * (h // 2)
    assert st.rays_shadow <= cst.rays_shadow and st.rays_reflect <= cst.rays_reflect


@pytest.mark.parametrize("name,w,h,t", [("cube", 320, 240, 8), ("bunny", 640, 360, 32), ("import", 480, 270, 32)])
def test_frame_supersampled_hash(name, w, h, t, scenes, gpu_ctx):
    scene, ctx = scenes(name), gpu_ctx(name)
    frame = scene.frame(w, h, threads=t, supersampling=True, rng_mode=ca.RNG_HASH, rng_seed=3)
    gpu_fb, st = ctx.render(frame)
    cpu_fb, _ = _oracle().render(scene.desc, frame, os_threads=8)
    _compare(gpu_fb, cpu_fb, "%s ss hash" % name)
    # another seed gives another image (the jitter is really applied)
    frame2 = scene.frame(w, h, threads=t, supersampling=True, rng_mode=ca.RNG_HASH, rng_seed=4)
    other, _ = ctx.render(frame2)
    assert (other != gpu_fb).any()


def test_frame_supersampled_libc_single_worker(scenes, gpu_ctx):
    """CTGPU_RNG_LIBC reproduces what ONE reference worker draws from glibc rand()."""
    scene, ctx = scenes("bunny"), gpu_ctx("bunny")
    frame = scene.frame(200, 160, threads=1, supersampling=True, rng_mode=ca.RNG_LIBC, rng_seed=1)
    gpu_fb, _ = ctx.render(frame)
    cpu_fb, _ = _oracle().render(scene.desc, frame, os_threads=1)
    _compare(gpu_fb, cpu_fb, "bunny ss libc T=1")
    # second frame continues the stream
    frame.rng_offset = 160 * 160 * 32
    gpu2, _ = ctx.render(frame)
    cpu2, _ = _oracle().render(scene.desc, frame, os_threads=1)
    _compare(gpu2, cpu2, "bunny ss libc T=1 frame 2")
    assert (gpu2 != gpu_fb).any()


@pytest.mark.parametrize("t,h", [(8, 240), (7, 240), (1, 121), (4, 66)])
def test_frame_subsampled(t, h, scenes, gpu_ctx):
    """settings.subsampling (the parser's default): every other row traced, rows between averaged."""
    scene, ctx = scenes("import"), gpu_ctx("import")
    for ss in (False, True):
        frame = scene.frame(320, h, threads=t, supersampling=ss, subsampling=True, rng_seed=5)
        gpu_fb, _ = ctx.render(frame)
        cpu_fb, _ = _oracle().render(scene.desc, frame, os_threads=4)
        _compare(gpu_fb, cpu_fb, "import subsample T=%d H=%d ss=%d" % (t, h, ss))
        assert ((gpu_fb == 0) == (cpu_fb == 0)).all()


@pytest.mark.parametrize("w,h", [(64, 2), (3, 64), (65, 63), (1000, 64), (64, 1000)])
def test_odd_bitmap_shapes(w, h, scenes, gpu_ctx):
    scene, ctx = scenes("cube"), gpu_ctx("cube")
    t = 1 if h < 8 else 8
    frame = scene.frame(w, h, threads=t, supersampling=False)
    gpu_fb, _ = ctx.render(frame)
    cpu_fb, _ = _oracle().render(scene.desc, frame, os_threads=2)
    _compare(gpu_fb, cpu_fb, "cube %dx%d" % (w, h))
    assert ((gpu_fb == 0) == (cpu_fb == 0)).all()


def test_rotated_camera(scenes, gpu_ctx):
    scene, ctx = scenes("import"), gpu_ctx("import")
    rot = _oracle().camera_rotation(np.float32(np.pi / 16), np.float32(np.pi / 8), np.float32(-np.pi / 16))
    frame = scene.frame(400, 300, threads=4, supersampling=False, rotation=rot)
    gpu_fb, _ = ctx.render(frame)
    cpu_fb, _ = _oracle().render(scene.desc, frame, os_threads=4)
    _compare(gpu_fb, cpu_fb, "import rotated")


def test_work_counters_match_oracle(scenes, gpu_ctx):
    """node-visit / triangle-test totals of the roofline model, against the oracle walking the
    same order with the same shadow early exit."""
    for name, w, h in (("bunny", 480, 270), ("cube", 160, 120)):
        scene, ctx = scenes(name), gpu_ctx(name)
        frame = scene.frame(w, h, threads=2, supersampling=False)
        got = ctx.count_work(frame)
        _, want = _oracle().render(scene.desc, frame, os_threads=4, shadow_early_exit=True, count_work=True)
        # the GPU skips the off-bitmap row y = -H/2 the reference traces and drops
        frame_rows_skipped = 1
        assert got.rays_shadow == want.rays_shadow - _rays_in_dropped_row(scene, frame, "shadow")
        assert got.node_visits == want.node_visits - _rays_in_dropped_row(scene, frame, "visits")
        assert got.triangle_tests == want.triangle_tests - _rays_in_dropped_row(scene, frame, "tests")
        assert frame_rows_skipped == 1


def _rays_in_dropped_row(scene, frame, what):
    """Work the oracle spends on the y = -H/2 row (band 0's first row), measured by rendering
    that band alone at T = rendered rows so that the band is exactly one row."""
    import copy
    f = copy.copy(frame)
    rows = (frame.height // frame.num_threads) * frame.num_threads
    if rows != frame.height:
        raise AssertionError("test assumes T divides H")
    f.num_threads = frame.height          # one row per band
    _, st = _oracle().render(scene.desc, f, os_threads=1, shadow_early_exit=True, count_work=True, band_first=0,
                             band_count=1)
    return {"shadow": st.rays_shadow, "visits": st.node_visits, "tests": st.triangle_tests}[what]


def test_partitions_assemble_to_the_same_frame(scenes, gpu_ctx):
    """Row tiles rendered as 1, 2, 3 and 8 partitions (all on this one GPU) and merged by
    ctgpu_assemble give the bit-identical framebuffer -- the property the multi-GPU path rests on."""
    torch = pytest.importorskip("torch")
    scene, ctx = scenes("bunny"), gpu_ctx("bunny")
    for ss, sub in ((False, False), (True, False), (False, True)):
        frame = scene.frame(640, 360, threads=8, supersampling=ss, subsampling=sub, rng_seed=9)
        whole, _ = ctx.render(frame)
        for parts, tile_rows in ((1, 8), (2, 8), (3, 16), (8, 8), (5, 1)):
            bufs = []
            n = ctx.partition_pixels(frame, ca.Partition(tile_rows, 0, parts))
            gathered = torch.zeros(parts * n, dtype=torch.int32, device="cuda")
            for p in range(parts):
                part = ca.Partition(tile_rows, p, parts)
                ctx.render_partition(frame, part, gathered[p * n:].data_ptr(), torch.cuda.current_stream().cuda_stream)
            fb = torch.zeros(frame.height * frame.width, dtype=torch.int32, device="cuda")
            ctx.assemble(frame, ca.Partition(tile_rows, 0, parts), gathered.data_ptr(), fb.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ctx.read_counters()
            got = fb.cpu().numpy().view(np.uint32).reshape(frame.height, frame.width)
            assert (got == whole).all(), "parts=%d tile_rows=%d ss=%d sub=%d" % (parts, tile_rows, ss, sub)
            bufs.append(got)
            # the host layer's assemble (tile buffers gathered into host memory) writes the same pixels
            on_host = np.zeros((frame.height, frame.width), dtype=np.uint32)
            tiles_host = np.ascontiguousarray(gathered.cpu().numpy().view(np.uint32))
            assert ca.host_library().cth_assemble_tiles(C.byref(frame), tile_rows, parts, tiles_host.ctypes.data,
                                                        on_host.ctypes.data) == 0
            assert (on_host == whole).all(), "host assemble: parts=%d tile_rows=%d ss=%d sub=%d" % (parts, tile_rows, ss, sub)


def test_native_rccl_frame_on_a_world_of_one(scenes, gpu_ctx):
    """The C++ side of the multi-GPU frame -- ctgpu_comm_init (ncclCommInitRank), ctgpu_render_sharded = trace my
    tiles + ncclGather to rank 0 + ct_assemble -- exercised with the only world a one-GPU box can form (RCCL does
    not let two ranks share a device): librccl is found and bound, the communicator comes up, the gather runs on
    the frame's stream, the assembled frame equals ctgpu_render's."""
    torch = pytest.importorskip("torch")
    scene = scenes("bunny")
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    try:
        ctx.comm_init(ca.GpuContext.comm_unique_id(), 0, 1)
        with pytest.raises(RuntimeError, match="already has a communicator"):
            ctx.comm_init(ca.GpuContext.comm_unique_id(), 0, 1)
        stream = torch.cuda.Stream()
        for ss, sub in ((False, False), (True, False), (False, True)):
            frame = scene.frame(640, 360, threads=8, supersampling=ss, subsampling=sub, rng_seed=9)
            want, _ = gpu_ctx("bunny").render(frame)
            fb = torch.zeros(frame.height * frame.width, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ctx.render_sharded(frame, 8, fb.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            ctx.read_counters()
            assert (fb.cpu().numpy().view(np.uint32).reshape(frame.height, frame.width) == want).all()
            # ctgpu_download_frame: exactly the pixels the reference writes reach the (pinned) host framebuffer
            host = torch.full((frame.height * frame.width,), 0x55AA55, dtype=torch.int32).pin_memory()
            ctx.download_frame(frame, fb.data_ptr(), host.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            got = host.numpy().view(np.uint32).reshape(frame.height, frame.width)
            written = want != 0
            assert (got[written] == want[written]).all()
            rows, cols = np.nonzero(written)
            box = np.zeros_like(written)
            box[rows.min():rows.max() + 1, cols.min():cols.max() + 1] = True
            assert (got[~box] == 0x55AA55).all()                   # nothing outside the rectangle was touched
    finally:
        ctx.close()
    # two contexts of one process (frames in flight) gather through ONE communicator (ctgpu_comm_share)
    a, b = ca.GpuContext(0), ca.GpuContext(0)
    try:
        for c in (a, b):
            c.upload(scene.desc)
        a.comm_init(ca.GpuContext.comm_unique_id(), 0, 1)
        b.comm_share(a)
        with pytest.raises(RuntimeError, match="already has a communicator"):
            b.comm_share(a)
        frame = scene.frame(640, 360, threads=8, supersampling=True, rng_seed=9)
        want, _ = gpu_ctx("bunny").render(frame)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        fbs = [torch.zeros(frame.height * frame.width, dtype=torch.int32, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        for rounds in range(3):
            for c, st_, fb in zip((a, b), streams, fbs):
                c.render_sharded(frame, 8, fb.data_ptr(), st_.cuda_stream)
        torch.cuda.synchronize()
        for c, fb in zip((a, b), fbs):
            c.read_counters()
            assert (fb.cpu().numpy().view(np.uint32).reshape(frame.height, frame.width) == want).all()
        assert a.comm_rank() == (0, 1) and b.comm_rank() == (0, 1)          # what the communicator itself reports
        # the order of the gathers: every sharded frame's gather carries {the context's place among the sharers, its frame
        # count}, and the root compares what arrives with its own.  A rank that issued its contexts' frames in another
        # order would send another pair -- forced here by stamping context b one frame off: the frame still assembles
        # (here, with one rank, even correctly), and the next counter read must refuse to vouch for it.
        b.comm_debug_skew(1)
        b.render_sharded(frame, 8, fbs[1].data_ptr(), streams[1].cuda_stream)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="DIFFERENT frames"):
            b.read_counters()
        b.comm_debug_skew(0)
        for c, st_, fb in zip((a, b), streams, fbs):
            c.render_sharded(frame, 8, fb.data_ptr(), st_.cuda_stream)
        torch.cuda.synchronize()
        a.read_counters(); b.read_counters()        # in step again: nothing to report
        a.close()                                  # the owner goes first: the communicator lives on with b
        b.render_sharded(frame, 8, fbs[1].data_ptr(), streams[1].cuda_stream)
        torch.cuda.synchronize()
        b.read_counters()
    finally:
        a.close()
        b.close()
    # no communicator: the sharded entry points say so instead of falling back to anything
    with pytest.raises(RuntimeError, match="no communicator"):
        gpu_ctx("bunny").render_sharded(scene.frame(64, 64, threads=1), 8, 0)


def test_measurement_hooks(scenes, monkeypatch):
    """ctgpu_kernel_timing / ctgpu_read_kernel_times and ctgpu_read_walk_work, what bench.py's roofline object is
    made of: one launch of each pipeline kernel per frame, event times that add up to the frame's kernel time,
    fetch counts of the two walks that are plausible (every packet fetches at least the root QuadNode).  The fetch
    counters are the shared walk's: the test asks for it."""
    scene = scenes("bunny")
    ctx = _context_with(monkeypatch, scene, CTGPU_PACKET=1, CTGPU_PACKET_BUDGET=None)
    try:
        frame = scene.frame(640, 360, threads=8, supersampling=True, rng_seed=1)
        ctx.render(frame)
        ctx.read_walk_work()
        ctx.kernel_timing(True)
        _, st = ctx.render(frame)
        _, st2 = ctx.render(frame)
        times = ctx.read_kernel_times()
        assert list(times.launches) == [2] * 5
        total = sum(times.total_ms)
        assert 0.7 * (st.kernel_ms + st2.kernel_ms) < total < 1.3 * (st.kernel_ms + st2.kernel_ms) + 0.1
        work = ctx.read_walk_work()
        n_primary = st.rays_primary + st2.rays_primary
        assert work.packets[0] * 64 >= n_primary and work.packets[0] * 64 < n_primary + 2 * 64 * 6144
        assert work.quad_nodes[0] >= work.packets[0] * 0.5 and work.triangle_records[0] > 0
        assert 0 < work.packets[1] * 64 < 2 * (st.rays_shadow + st2.rays_shadow) + 2 * 64 * 6144
        assert work.quad_nodes[1] > 0 and work.triangle_records[1] > 0
        ctx.kernel_timing(False)
        ctx.render(frame)
        assert list(ctx.read_kernel_times().launches) == [0] * 5
    finally:
        ctx.close()


def test_dragon_full_size_properties(scenes, gpu_ctx, monkeypatch):
    """BASELINE.json's headline workload at full size (the oracle would take minutes here):
    size-independent properties -- determinism, ray accounting, tile-split invariance against
    a band of oracle rows."""
    scene, ctx = scenes("dragon"), gpu_ctx("dragon")
    frame = scene.frame(1920, 1080, supersampling=False)          # T = 32 from the scene file
    a, sa = ctx.render(frame)
    b, sb = ctx.render(frame)
    assert (a == b).all() and sa.rays == sb.rays                  # idempotent
    hits = int(((a != 0x333333) & (a != 0)).sum())
    # two non-ambient lights per hit: traced, or left out because the light cannot contribute there
    assert sa.rays_shadow + sa.rays_shadow_elided == 2 * hits and 0 <= sa.rays_shadow_elided < hits * 2
    lean = _context_with(monkeypatch, scene, CTGPU_KEEP_SHADOW_RAYS=None)          # (the default, whatever the suite runs under)
    try:
        c, sc = lean.render(frame)
        assert (c == a).all() and sc.rays_shadow + sc.rays_shadow_elided == 2 * hits and sc.rays_shadow_elided > 0
    finally:
        lean.close()
    assert sa.rays_primary == 1080 * 1056 - 1080                  # rendered rows x square, minus the dropped row
    assert (a[1:25] == 0).all() and (a[0] == 0).all() and (a[:, :420] == 0).all() and (a[:, 1500:] == 0).all()
    # four bands (of 32) against the oracle, bit for bit
    cpu_fb, _ = _oracle().render(scene.desc, frame, os_threads=8, band_first=14, band_count=4)
    rows = cpu_fb.any(axis=1)
    assert rows.sum() == 4 * 33
    _compare(a[rows], cpu_fb[rows], "dragon stand-in 1920x1080 bands 14-17")


def _oracle_bands(scene, frame, bands):
    """Rows of the frame the oracle renders for the given row bands (worker partitions,
    raythread.cpp:574-588), one band at a time so that any set of bands can be picked."""
    rows = np.zeros(frame.height, dtype=bool)
    fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
    for b in bands:
        part, _ = _oracle().render(scene.desc, frame, os_threads=1, band_first=b, band_count=1)
        band_rows = part.any(axis=1)
        fb[band_rows] = part[band_rows]
        rows |= band_rows
    return fb, rows


def _oracle_bands_parallel(scene, frame, bands):
    """The same with the bands traced concurrently (one oracle worker thread per band)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(bands), 8)) as pool:
        parts = list(pool.map(lambda b: _oracle_bands(scene, frame, [b]), bands))
    fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
    rows = np.zeros(frame.height, dtype=bool)
    for part, r in parts:
        fb[r] = part[r]
        rows |= r
    return fb, rows


def test_dragon_headline_frame_supersampled(scenes, gpu_ctx):
    """THE benchmarked instantiation (bench.py's default: BASELINE.json's metric configuration):
    scene_import_dragon.json at 1920x1080, the scene file's own settings -- 32 bands, supersampling on
    (16 jittered samples per pixel, hash jitter seed 0), three lights with shadow rays.  Six of the 32
    row bands, spread over the image, against the oracle; plus whole-frame properties."""
    scene, ctx = scenes("dragon"), gpu_ctx("dragon")
    frame = scene.frame(1920, 1080, rng_mode=ca.RNG_HASH, rng_seed=0)          # T = 32, ss = on from the scene file
    assert frame.flags & ca.F_SUPERSAMPLE and frame.num_threads == 32
    a, sa = ctx.render(frame)
    b, sb = ctx.render(frame)
    assert (a == b).all() and sa.rays == sb.rays                               # idempotent
    assert sa.rays_primary == (1080 * 1056 - 1080) * 16
    assert (a[1:25] == 0).all() and (a[0] == 0).all() and (a[:, :420] == 0).all() and (a[:, 1500:] == 0).all()
    bands = [3, 9, 14, 17, 22, 28]
    cpu_fb, rows = _oracle_bands_parallel(scene, frame, bands)
    assert rows.sum() == len(bands) * 33
    n_diff = _compare(a[rows], cpu_fb[rows], "dragon stand-in 1920x1080 ss=on, bands %s" % bands)
    assert n_diff == 0            # observed: bit-identical (the 1-LSB allowance is for pow(), which has not bitten yet)
    assert ((a[rows] != 0x333333) & (a[rows] != 0)).sum() > 50_000             # the bands show the mesh


@pytest.mark.parametrize("name,bands,min_hit", [("import", [2, 9, 15, 20, 27], 60_000), ("bunny", [4, 10, 16, 23, 29], 40_000)])
def test_c2_c3_scene_default_frames_at_full_size(name, bands, min_hit, scenes, gpu_ctx):
    """BASELINE.json configs[1] and [2] (C2 scene_import.json, C3 scene_import_bunny.json) at 1920x1080 with the scene
    files' OWN settings -- 32 bands, supersampling on (16 samples per pixel, hash jitter seed 0), three lights with
    shadow rays: five of the 32 row bands, spread over the image, against the oracle, bit for bit; whole-frame
    properties (idempotence, ray accounting, the untouched border)."""
    scene, ctx = scenes(name), gpu_ctx(name)
    frame = scene.frame(1920, 1080, rng_mode=ca.RNG_HASH, rng_seed=0)
    assert frame.flags & ca.F_SUPERSAMPLE and not (frame.flags & ca.F_SUBSAMPLE) and frame.num_threads == 32
    a, sa = ctx.render(frame)
    b, sb = ctx.render(frame)
    assert (a == b).all() and sa.rays == sb.rays
    assert sa.rays_primary == (1080 * 1056 - 1080) * 16
    assert (a[1:25] == 0).all() and (a[0] == 0).all() and (a[:, :420] == 0).all() and (a[:, 1500:] == 0).all()
    cpu_fb, rows = _oracle_bands_parallel(scene, frame, bands)
    assert rows.sum() == len(bands) * 33
    n_diff = _compare(a[rows], cpu_fb[rows], "%s 1920x1080 ss=on (scene default), bands %s" % (name, bands))
    assert n_diff == 0
    assert ((a[rows] != 0x333333) & (a[rows] != 0)).sum() > min_hit


def test_dragon_4k_supersampled_c5(scenes, gpu_ctx):
    """BASELINE.json configs[4] (C5): dragon at 3840x2160, 16 samples per pixel and shadow rays --
    74 M primary rays.  Four of the 32 bands against the oracle, whole-frame ray accounting."""
    scene, ctx = scenes("dragon"), gpu_ctx("dragon")
    frame = scene.frame(3840, 2160, rng_mode=ca.RNG_HASH, rng_seed=0)
    assert frame.flags & ca.F_SUPERSAMPLE and frame.num_threads == 32
    a, sa = ctx.render(frame)
    assert sa.rays_primary == (2160 * 2144 - 2160) * 16                        # 2160 columns x 2144 rows, minus the dropped row
    assert (a[:, :840] == 0).all() and (a[:, 3000:] == 0).all() and (a[0] == 0).all()
    bands = [6, 13, 18, 25]
    cpu_fb, rows = _oracle_bands_parallel(scene, frame, bands)
    assert rows.sum() == len(bands) * 67
    n_diff = _compare(a[rows], cpu_fb[rows], "dragon stand-in 3840x2160 ss=on (C5), bands %s" % bands)
    assert n_diff == 0
    assert ((a[rows] != 0x333333) & (a[rows] != 0)).sum() > 100_000


def test_both_gpu_paths_agree(scenes, monkeypatch):
    """The wavefront pipeline (fast path) and the megakernel (general path) are two independent
    implementations of the same semantics: bit-identical frames, identical ray counts."""
    scene = scenes("bunny")
    monkeypatch.setenv("CTGPU_FORCE_MEGAKERNEL", "1")
    mega = ca.GpuContext(0)
    monkeypatch.delenv("CTGPU_FORCE_MEGAKERNEL")
    fast = ca.GpuContext(0)
    mega.upload(scene.desc)
    fast.upload(scene.desc)
    try:
        for ss, sub, w, h in ((False, False, 1920, 1080), (True, False, 960, 540), (False, True, 640, 360),
                              (True, True, 320, 250)):
            frame = scene.frame(w, h, supersampling=ss, subsampling=sub, rng_seed=21)
            a, sa = mega.render(frame)
            b, sb = fast.render(frame)
            assert (a == b).all()
            # (the megakernel traces every shadow ray the reference does; the pipeline leaves out those
            # whose answer cannot matter and says how many)
            assert (sa.rays_primary, sa.rays_shadow) == (sb.rays_primary, sb.rays_shadow + sb.rays_shadow_elided)
    finally:
        mega.close()
        fast.close()


@pytest.mark.parametrize("env", [{"CTGPU_LDS_DEPTH": "1"}, {"CTGPU_LDS_DEPTH": "3", "CTGPU_CHUNK": "64"},
                                 {"CTGPU_CHUNK": "4096"}, {"CTGPU_WALK_THRESH": "8,16,8"},
                                 {"CTGPU_BLOCKS_PER_CU": "1"}, {"CTGPU_KEEP_SHADOW_RAYS": "1"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_pipeline_knobs_do_not_change_pixels(env, scenes, gpu_ctx, monkeypatch):
    """Stack spill to HBM (tiny LDS depth), chunk sizes, phase thresholds and the
    persistent grid size are performance knobs only: every setting renders the identical frame."""
    scene = scenes("bunny")
    frames = [scene.frame(640, 360, supersampling=False), scene.frame(320, 200, threads=8, supersampling=True, rng_seed=5)]
    want = [gpu_ctx("bunny").render(f)[0] for f in frames]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = ca.GpuContext(0)
    for k in env:
        monkeypatch.delenv(k)
    ctx.upload(scene.desc)
    try:
        for f, w in zip(frames, want):
            got, _ = ctx.render(f)
            assert (got == w).all()
    finally:
        ctx.close()


def test_elided_shadow_rays_are_accounted_for(scenes, gpu_ctx, monkeypatch):
    """ct_light_gate leaves out the shadow rays whose answer ComputeLighting never uses (n.l <= 0 and
    no specular reach, raythread.cpp:309-325).  With CTGPU_KEEP_SHADOW_RAYS=1 every one is traced:
    same pixels, and traced + elided is the same number either way -- the reference's ray count."""
    monkeypatch.setenv("CTGPU_KEEP_SHADOW_RAYS", "1")
    ctxs = {}
    try:
        for name in ("bunny", "import"):
            ctxs[name] = ca.GpuContext(0)
            ctxs[name].upload(scenes(name).desc)
        monkeypatch.delenv("CTGPU_KEEP_SHADOW_RAYS")
        leans = {name: _context_with(monkeypatch, scenes(name), CTGPU_KEEP_SHADOW_RAYS=None) for name in ("bunny", "import")}
        ctxs.update({"lean " + k: v for k, v in leans.items()})
        for name, w, h in (("bunny", 640, 360), ("import", 480, 270)):
            scene = scenes(name)
            for ss in (False, True):
                frame = scene.frame(w, h, threads=8, supersampling=ss, rng_seed=6)
                lean, sl = leans[name].render(frame)
                full, sf = ctxs[name].render(frame)
                assert (lean == full).all()
                assert sf.rays_shadow_elided == 0 and sl.rays_shadow_elided > 0
                assert sl.rays_shadow + sl.rays_shadow_elided == sf.rays_shadow
                if not ss:
                    # ... which is what the oracle traces, less the row the bitmap drops (not traced here)
                    _, cst = _oracle().render(scene.desc, frame, os_threads=4)
                    import copy
                    f1 = copy.copy(frame)
                    f1.num_threads = frame.height
                    _, dropped = _oracle().render(scene.desc, f1, os_threads=1, band_first=0, band_count=1)
                    assert sf.rays_shadow == cst.rays_shadow - dropped.rays_shadow
    finally:
        for c in ctxs.values():
            c.close()


def _write_scene(tmp_path, name, triangles, lights, settings=None, camera=(0, 0, -3)):
    import json
    objs = [{"type": "triangle", "p1": t[0], "p2": t[1], "p3": t[2], "color": list(t[3]), "specular": t[4],
             "reflection": 0} for t in triangles]
    scene = {"objects": objs, "camera": {"position": list(camera)},
             "settings": settings or {"numberOfThreads": 4, "subsampling": False, "wireframe": False,
                                      "supersampling": False}}
    if lights:       # an empty list is a syntax error for the reference's reader (scenefile.cpp:142), and ours
        scene["lights"] = lights
    path = tmp_path / name
    path.write_text(json.dumps(scene))
    return str(path)


def _render_both(path, w, h, **kw):
    scene = ca.HostScene(path)
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    try:
        frame = scene.frame(w, h, **kw)
        gpu_fb, st = ctx.render(frame)
        cpu_fb, cst = _oracle().render(scene.desc, frame, os_threads=2)
        return gpu_fb, cpu_fb, st, cst
    finally:
        ctx.close()


def test_unusual_scenes(tmp_path):
    """One-triangle tree (the root is a leaf), no lights at all, ambient only, more shadow-casting
    lights than the fast path's 32 (takes the megakernel), specular -1, axis-aligned flat geometry
    that lands box planes exactly on hit distances (exercises the uncertain-verdict paths)."""
    tri = lambda z, c, s: ([-1.5, -1, z], [1.5, -1, z], [0, 1.5, z], c, s)
    ambient = {"type": "ambient", "intensity": 0.3}
    point = lambda x: {"type": "point", "intensity": 0.02, "position": [x, 2, -4]}
    cases = {
        "one_triangle": ([tri(2, (255, 0, 0), 50)], [ambient, point(1)]),
        "no_lights": ([tri(2, (0, 255, 0), 50), tri(3, (0, 0, 255), 10)], []),
        "ambient_only": ([tri(2, (10, 200, 90), -1), tri(4, (200, 10, 90), 5)], [ambient]),
        "forty_lights": ([tri(2, (250, 250, 250), 20), tri(1, (40, 90, 250), -1), ([-4, -1.2, 0], [4, -1.2, 0], [0, -1.2, 9], (200, 200, 60), 100)],
                         [ambient] + [point(-4 + 0.2 * i) for i in range(40)]),
        "flat_axis_aligned": ([([-2, -1, 1], [2, -1, 1], [2, -1, 5], (255, 255, 255), 100),
                               ([-2, -1, 1], [2, -1, 5], [-2, -1, 5], (255, 255, 255), 100),
                               ([-1, -1, 3], [1, -1, 3], [0, 1, 3], (255, 80, 80), 10),
                               ([-2, -1, 5], [2, -1, 5], [2, 2, 5], (80, 80, 255), 10),
                               ([-2, -1, 5], [2, 2, 5], [-2, 2, 5], (80, 80, 255), 10)],
                              [ambient, {"type": "point", "intensity": 0.6, "position": [0, 3, 3]},
                               {"type": "directional", "intensity": 0.2, "direction": [0, 1, 0]}]),
    }
    for name, (tris, lights) in cases.items():
        path = _write_scene(tmp_path, name + ".json", tris, lights)
        for ss in (False, True):
            gpu_fb, cpu_fb, st, cst = _render_both(path, 200, 160, threads=4, supersampling=ss, rng_seed=1)
            _compare(gpu_fb, cpu_fb, "%s ss=%d" % (name, ss))
            assert ((gpu_fb == 0) == (cpu_fb == 0)).all()
            assert (gpu_fb != 0x333333).any() or name == "never"


def test_empty_and_invalid_uploads():
    ctx = ca.GpuContext(0)
    try:
        d = ca.SceneDesc()
        with pytest.raises(RuntimeError, match="empty scene"):
            ctx.upload(d)
        scene = ca.HostScene(vertices=np.array([[0, 0, 2, 1, 0, 2, 0, 1, 2.0]]))
        with pytest.raises(RuntimeError, match="no scene"):
            ctx.render(scene.frame(64, 64, threads=1, supersampling=False))
        ctx.upload(scene.desc)
        bad = scene.frame(64, 64, threads=1, supersampling=False)
        bad.num_threads = 0
        with pytest.raises(RuntimeError, match="num_threads"):
            ctx.render(bad)
        bad = scene.frame(64, 64, threads=1, supersampling=False)
        bad.max_depth = 11
        with pytest.raises(RuntimeError, match="max_depth"):
            ctx.render(bad)
        # corrupt tree: child link out of range is rejected on the host, never reaches the device
        nodes = (ca.Node * 1)()
        nodes[0].count = 0
        nodes[0].left = 7
        d = scene.desc
        broken = ca.SceneDesc(1, 1, 0, 0, d.vertices, d.materials, nodes, d.indexes, d.lights)
        with pytest.raises(RuntimeError, match="child link"):
            ctx.upload(broken)
    finally:
        ctx.close()


def test_single_leaf_tree_on_every_path(monkeypatch):
    """A scene whose BVH is one leaf: the root reference is a leaf reference.  Exercised on the
    wavefront pipeline, the megakernel and the ray-list hook."""
    verts = np.array([[-1.5, -1, 2, 1.5, -1, 2, 0, 1.5, 2.0], [-1.5, -1, 3, 1.5, -1, 3, 0, 1.5, 3.0]])
    scene = ca.HostScene(vertices=verts)
    assert scene.desc.n_nodes == 1
    frame = scene.frame(128, 96, threads=2, supersampling=False)
    frame.cam_pos = (C.c_double * 3)(0, 0, -3)
    want, _ = _oracle().render(scene.desc, frame, os_threads=1)
    for force in ("0", "1"):
        monkeypatch.setenv("CTGPU_FORCE_MEGAKERNEL", force)
        ctx = ca.GpuContext(0)
        ctx.upload(scene.desc)
        try:
            got, _ = ctx.render(frame)
            assert (got == want).all()
            rays = np.zeros(1000, dtype=ca.bindings.RAY_DTYPE)
            rng = np.random.default_rng(3)
            rays["origin"] = [0, 0, -3]
            rays["direction"] = np.c_[rng.normal(size=(1000, 2)) * 0.4, np.ones(1000)]
            rays["t"] = np.float32(1e30)
            hits, ref = ctx.trace_rays(rays), _oracle().trace_rays(scene.desc, rays)
            assert (hits["triangle"] == ref["triangle"]).all() and (hits["found"] == ref["found"]).all()
            assert (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all() and ref["found"].sum() > 100
        finally:
            ctx.close()
    monkeypatch.delenv("CTGPU_FORCE_MEGAKERNEL")


def test_packet_walk_variant(scenes, gpu_ctx, monkeypatch):
    """ct_walk_packet (the wave's 64 rays share one walk over the four-wide folded tree, the default)
    and the per-lane ct_walk over the reference's binary tree render identical frames."""
    monkeypatch.setenv("CTGPU_PACKET", "0")
    ctxs = {}
    try:
        for name in ("bunny", "import"):
            ctxs[name] = ca.GpuContext(0)
            ctxs[name].upload(scenes(name).desc)
        monkeypatch.delenv("CTGPU_PACKET")
        for name, w, h, t in (("bunny", 640, 360, 32), ("import", 480, 270, 8), ("bunny", 200, 160, 1)):
            scene = scenes(name)
            for ss, sub in ((True, False), (True, True), (False, False), (False, True)):
                frame = scene.frame(w, h, threads=t, supersampling=ss, subsampling=sub, rng_seed=5)
                want, sw = gpu_ctx(name).render(frame)
                got, sg = ctxs[name].render(frame)
                assert (got == want).all(), (name, w, h, ss, sub)
                assert sg.rays == sw.rays
                assert sg.walk_kind == ca.bindings.WALK_PER_LANE and not sg.shadow_tree and not sg.cull_tree
    finally:
        for c in ctxs.values():
            c.close()


def _tree_with_boxes_edited(scene, edit):
    """A copy of the scene description whose node boxes went through edit(nodes) -- the caller of
    the C ABI may hand over any tree, not only one built by BuildBVH."""
    d = scene.desc
    nodes = (ca.Node * d.n_nodes)()
    C.memmove(nodes, d.nodes, C.sizeof(ca.Node) * d.n_nodes)
    edit(nodes, d.n_nodes)
    copy = ca.SceneDesc(d.n_triangles, d.n_nodes, d.n_lights, 0, d.vertices, d.materials, nodes, d.indexes, d.lights)
    copy._keep = nodes
    return copy


def test_folded_tree_when_boxes_do_not_nest(scenes):
    """The packet walk folds every second level of the tree away, which is exact only where a
    node's children's boxes lie inside its own; the upload checks that per node.  Here some inner
    nodes are shrunk below their children and some children are inflated beyond their parent: those
    nodes must stay unfolded, and every path still takes exactly the reference's decisions for the
    tree AS GIVEN (the oracle walks the same edited tree)."""
    scene = scenes("import")

    def shrink_and_inflate(nodes, n):
        rng = np.random.default_rng(11)
        inner = [i for i in range(1, n) if nodes[i].count == 0]
        for i in rng.choice(inner, size=max(len(inner) // 5, 1), replace=False):
            for a in range(3):
                lo, hi = nodes[i].bmin[a], nodes[i].bmax[a]
                if rng.random() < 0.5:                       # shrink: children now stick out
                    nodes[i].bmin[a] = lo + 0.25 * (hi - lo)
                    nodes[i].bmax[a] = hi - 0.25 * (hi - lo)
                else:                                        # inflate a child beyond this node
                    c = nodes[i].left + int(rng.integers(2))
                    nodes[c].bmax[a] = nodes[c].bmax[a] + 0.5 * (hi - lo) + 0.01

    desc = _tree_with_boxes_edited(scene, shrink_and_inflate)
    ctx = ca.GpuContext(0)
    ctx.upload(desc)
    try:
        for ss, w, h in ((False, 480, 270), (True, 240, 136)):
            frame = scene.frame(w, h, threads=8, supersampling=ss, rng_seed=3)
            got, st = ctx.render(frame)
            want, cst = _oracle().render(desc, frame, os_threads=2)
            _compare(got, want, "edited tree ss=%d" % ss)
            # the edit changes what the reference itself renders: make sure it bit
            plain, _ = _oracle().render(scene.desc, frame, os_threads=2)
            assert (plain != want).any()
        rays = _random_rays(scene, 20000, 5)
        hits, ref = ctx.trace_rays(rays), _oracle().trace_rays(desc, rays)
        assert (hits["triangle"] == ref["triangle"]).all() and (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    finally:
        ctx.close()


def test_filter_switches_off_for_huge_coordinates(scenes):
    """The binary32 plane filter needs a finite bound on the planes of the tree; with a box plane at
    1e35 there is none (and that plane overflows binary32): every box verdict then comes from the
    binary64 / exact levels, and the frame is still the reference's."""
    scene = scenes("import")

    def blow_up(nodes, n):
        nodes[0].bmax[1] = 1e35
        leaf = next(i for i in range(n - 1, 0, -1) if nodes[i].count)
        nodes[leaf].bmax[1] = 1e35
        nodes[leaf].bmin[0] = -1e35

    desc = _tree_with_boxes_edited(scene, blow_up)
    ctx = ca.GpuContext(0)
    ctx.upload(desc)
    try:
        for ss, w, h in ((False, 320, 180), (True, 160, 90)):
            frame = scene.frame(w, h, threads=4, supersampling=ss, rng_seed=2)
            got, _ = ctx.render(frame)
            want, _ = _oracle().render(desc, frame, os_threads=2)
            _compare(got, want, "huge plane ss=%d" % ss)
            assert (got != 0x333333).any()
    finally:
        ctx.close()


def test_rays_with_nan_box_parameters(tmp_path):
    """Camera exactly on the plane of flat, axis-aligned geometry, looking along it: the centre row of
    a frame without supersampling has d.y == 0 and o.y == a box plane, so box parameters are 0/0 = NaN
    and the reference's min / max macros do whatever they do with them.  The folded tree is not
    proven for such rays: the packet walk must notice and walk them again unfolded.  Every path has
    to reproduce the reference's (odd) picture, ray for ray."""
    floor, extra = [], []
    for i in range(8):                       # a tiled floor in the plane y = -1 (flat boxes all the way down the tree)
        for j in range(8):
            x, z = -2 + 0.5 * i, 1 + 0.5 * j
            floor.append(([x, -1, z], [x + 0.5, -1, z], [x + 0.5, -1, z + 0.5], (255, 255, 255), 100))
            floor.append(([x, -1, z], [x + 0.5, -1, z + 0.5], [x, -1, z + 0.5], (230, 230, 230), 100))
    for i in range(6):                       # and things standing on it
        x = -1.75 + 0.6 * i
        extra.append(([x, -1, 3], [x + 0.4, -1, 3], [x + 0.2, 0.5, 3], (255, 80, 80), 10))
        extra.append(([x, -1, 4.5], [x + 0.4, -1, 4.5], [x + 0.2, 1.0, 4.5], (80, 80, 255), 10))
    lights = [{"type": "ambient", "intensity": 0.3}, {"type": "point", "intensity": 0.6, "position": [0, 3, 3]},
              {"type": "directional", "intensity": 0.2, "direction": [0, 1, 0]}]
    path = _write_scene(tmp_path, "on_the_floor.json", floor + extra, lights, camera=(0, -1, -3))
    scene = ca.HostScene(path)
    assert scene.desc.n_nodes > 30
    rays = np.zeros(4000, dtype=ca.bindings.RAY_DTYPE)
    rng = np.random.default_rng(9)
    rays["origin"] = [0, -1, -3]
    rays["direction"] = np.c_[rng.normal(size=4000) * 0.5, np.zeros(4000), np.ones(4000)]
    rays["t"] = np.float32(1e30)
    for packet in ("1", "0"):
        with _knob("CTGPU_PACKET", packet):
            ctx = ca.GpuContext(0)
        ctx.upload(scene.desc)
        try:
            for ss, sub in ((False, False), (True, False), (False, True)):
                frame = scene.frame(200, 160, threads=4, supersampling=ss, subsampling=sub, rng_seed=1)
                got, st = ctx.render(frame)
                want, cst = _oracle().render(scene.desc, frame, os_threads=2)
                _compare(got, want, "camera on the floor plane, packet=%s ss=%d sub=%d" % (packet, ss, sub))
            hits, ref = ctx.trace_rays(rays), _oracle().trace_rays(scene.desc, rays)
            assert (hits["triangle"] == ref["triangle"]).all() and (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
            # the same rays through the frame kernel's list mode, also starting at ray.t = 0 (a reflected ray's start,
            # raythread.cpp:373): a ray that leaves the packet for the unfolded second pass restarts from ITS start value
            for t_start in (1e30, 0.0):
                listed = rays.copy()
                listed["t"] = np.float32(t_start)
                hits, ref = ctx.trace_rays_packet(listed), _oracle().trace_rays(scene.desc, listed)
                assert (hits["triangle"] == ref["triangle"]).all() and (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
                assert (hits["tclosest"].view(np.uint32) == ref["tclosest"].view(np.uint32)).all(), t_start
        finally:
            ctx.close()


def _tables_scene(tmp_path):
    """Flat-topped tables whose tops lie in the plane y = 0.5 and whose walls end there (box MAX planes
    at y = 0.5 all the way down the tree), rows of tall triangles that cross that plane between and
    behind the rows of tables (inner nodes hold both, deeper ones only tables), some walls ending
    exactly at x = 0."""
    tris = []
    for j in range(4):
        z0 = 1.0 + 1.3 * j
        for i in range(7):
            x0 = round((i - 3) * 0.4 - 0.4, 6)
            x1, z1 = round(x0 + 0.3 + 0.1 * (i == 2), 6), z0 + 0.3
            wall, top = (200, 200, 60 + 40 * j), (90, 160, 220)
            tris.append(([x0, -1, z0], [x1, -1, z0], [x1, 0.5, z0], wall, 30))            # front wall, ends at y = 0.5
            tris.append(([x0, -1, z0], [x1, 0.5, z0], [x0, 0.5, z0], wall, 30))
            tris.append(([x0, 0.5, z0], [x1, 0.5, z0], [x1, 0.5, z1], top, 100))           # flat top in y = 0.5
            tris.append(([x0, 0.5, z0], [x1, 0.5, z1], [x0, 0.5, z1], top, 100))
        if j % 2:                                                                          # a row of tall ones behind
            for i in range(9):
                x = -1.9 + 0.42 * i
                tris.append(([x, -1, z0 + 0.8], [x + 0.4, -1, z0 + 0.8], [x + 0.2, 2.0 - 0.1 * i, z0 + 0.8],
                             (255, 70 + 20 * i, 70), 10))                                  # crosses y = 0.5
    lights = [{"type": "ambient", "intensity": 0.3}, {"type": "point", "intensity": 0.6, "position": [0.3, 3, 0]},
              {"type": "directional", "intensity": 0.2, "direction": [0.2, 1, -0.3]}]
    return _write_scene(tmp_path, "tables.json", tris, lights, camera=(0, 0.5, -3))


def test_nan_parameter_rays_that_hit(tmp_path):
    """Camera height exactly on the MAX plane of many boxes: for the centre row (d.y == 0) that plane's
    parameter is 0/0.  The reference's min / max macros (mymath.h:11-17) then DROP the x and y
    constraints of such a box (bvh.cpp:165-179 evaluated in order), the ray enters it on its z range
    alone and goes on to hit the tall triangles behind.  The packet walk must hand such rays to its
    unfolded second pass and walk them there to the END (round 1 truncated that walk: a shadowed
    variable let pass 1 give rays up again).  Frames and single rays, every path, against the oracle."""
    scene = ca.HostScene(_tables_scene(tmp_path))
    assert scene.desc.n_nodes > 30
    w, h = 240, 160
    rays = np.zeros(6400, dtype=ca.bindings.RAY_DTYPE)
    rng = np.random.default_rng(4)
    rays["origin"] = [0, 0.5, -3]
    rays["direction"] = np.c_[rng.normal(size=6400) * 0.45, np.zeros(6400), np.ones(6400)]
    rays["direction"][:640, 0] = 0.0                  # ... and a zero x component with walls ending at x = 0
    rays["direction"][320:640, 1] = rng.normal(size=320) * 0.2
    rays["t"] = np.float32(1e30)
    ref = _oracle().trace_rays(scene.desc, rays)
    assert ref["found"][640:].sum() > 1500            # the in-plane rays really hit (the tall triangles)
    for packet in ("1", "0"):
        with _knob("CTGPU_PACKET", packet):
            ctx = ca.GpuContext(0)
        ctx.upload(scene.desc)
        try:
            for ss, sub in ((False, False), (True, False), (False, True)):
                frame = scene.frame(w, h, threads=4, supersampling=ss, subsampling=sub, rng_seed=2)
                got, st = ctx.render(frame)
                want, cst = _oracle().render(scene.desc, frame, os_threads=2)
                if not ss:
                    centre = want[h // 2, w // 2 - h // 2: w // 2 + h // 2]
                    assert ((centre != 0x333333) & (centre != 0)).sum() > h // 4       # the d.y == 0 row shows geometry
                _compare(got, want, "camera on box max planes, packet=%s ss=%d sub=%d" % (packet, ss, sub))
                assert (got == want).all()
            for hits in (ctx.trace_rays(rays), ctx.trace_rays_packet(rays)):
                assert (hits["found"] == ref["found"]).all()
                assert (hits["triangle"] == ref["triangle"]).all()
                assert (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
            assert (ctx.trace_rays_packet(rays, any_hit=True)["found"] == ref["found"]).all()
        finally:
            ctx.close()


@pytest.mark.parametrize("name", ["cube", "import", "bunny"])
def test_closest_hit_exact_through_the_packet_walk(name, scenes, gpu_ctx):
    """north_star: intermediate float hits within 1e-4 rel, exact index -- here bit-exact t, tclosest and
    index, on the kernel that renders the frames (ct_walk_packet), 64 list entries per packet: random
    incoherent rays (every lane its own way through the shared walk), rays with zero direction
    components (exact level, NaN / inf slabs), reflected-ray starts (t = 0, raythread.cpp:373), and
    coherent camera-like bundles (where the binary32 filter decides nearly everything)."""
    scene, ctx = scenes(name), gpu_ctx(name)
    n = 200_000 if name != "cube" else 50_000
    rays = _random_rays(scene, n, seed=12)
    v = scene.vertices().reshape(-1, 3)
    centre, span = (v.min(0) + v.max(0)) / 2, (v.max(0) - v.min(0)).max()
    rng = np.random.default_rng(5)
    bundle = np.zeros(n // 2, dtype=ca.bindings.RAY_DTYPE)
    eye = centre + np.array([0.1, 0.2, -2.5]) * span
    bundle["origin"] = eye
    side = int(np.sqrt(len(bundle)))
    gx, gy = np.meshgrid(np.linspace(-0.4, 0.4, side), np.linspace(-0.4, 0.4, side))
    k = side * side
    bundle["direction"][:k] = np.c_[gx.ravel(), gy.ravel(), np.ones(k)] * 0.7 + (centre - eye) / span * 0.3
    bundle["direction"][k:] = rng.normal(size=(len(bundle) - k, 3))
    bundle["t"] = np.float32(1e30)
    for what, batch in (("random", rays), ("bundle", bundle), ("odd length", rays[:1000 + 37])):
        got = ctx.trace_rays_packet(batch)
        want = _oracle().trace_rays(scene.desc, batch)
        assert (got["found"] == want["found"]).all(), what
        assert (got["triangle"] == want["triangle"]).all(), what
        assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all(), what
        assert (got["tclosest"].view(np.uint32) == want["tclosest"].view(np.uint32)).all(), what
        assert want["found"].sum() > len(batch) // 20, what
        # the shadow query of the same rays: hit or not is all ComputeLighting asks (raythread.cpp:303-307)
        fresh = batch.copy()
        fresh["t"] = np.float32(1e30)
        shadowed = ctx.trace_rays_packet(fresh, any_hit=True)
        assert (shadowed["found"] == _oracle().trace_rays(scene.desc, fresh)["found"]).all(), what


@pytest.mark.parametrize("name", ["import", "bunny"])
def test_rays_with_zero_direction_components(name, scenes, gpu_ctx):
    """Axis-parallel and plane-parallel rays (one or two direction components +0 or -0): the reference's box
    parameters on such an axis are infinities -- or NaNs when the origin lies exactly on a box plane.  The packet
    walk decides these from the binary32 planes where it can (origin strictly inside / outside), from the
    binary64 boxes where the roundings coincide, and by the reference's own divisions for the NaNs.  Origins:
    random; exactly on vertex coordinates (= planes of leaf boxes: NaNs); one binary64 step beside them (same
    binary32 value: the filter cannot tell the side); whole packets of them and single ones among ordinary rays."""
    scene, ctx = scenes(name), gpu_ctx(name)
    rng = np.random.default_rng(77)
    v = scene.vertices().reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    n = 120_000
    rays = np.zeros(n, dtype=ca.bindings.RAY_DTYPE)
    origin = lo + rng.random((n, 3)) * (hi - lo)                  # inside the mesh's bound: inside many boxes
    origin[: n // 3] += rng.normal(size=(n // 3, 3)) * (hi - lo)  # and outside
    target = lo + rng.random((n, 3)) * (hi - lo)
    direction = target - origin
    axes = rng.integers(0, 3, n)
    zero = np.where(rng.random(n) < 0.5, 0.0, -0.0)
    direction[np.arange(n), axes] = zero                          # one zero component, +0 or -0
    second = rng.random(n) < 0.2
    direction[second, (axes[second] + 1) % 3] = 0.0               # a fifth: two of them
    # origins on planes: a vertex coordinate on the ray's zero axis (and sometimes on another axis too)
    on_plane = rng.random(n) < 0.4
    pick = v[rng.integers(0, len(v), n)]
    origin[on_plane, axes[on_plane]] = pick[on_plane, axes[on_plane]]
    beside = on_plane & (rng.random(n) < 0.5)                     # ... or the neighbouring double
    origin[beside, axes[beside]] = np.nextafter(origin[beside, axes[beside]], np.where(rng.random(beside.sum()) < 0.5, -np.inf, np.inf))
    ordinary = rng.random(n) < 0.3                                # ordinary rays among them (the tail of the list)
    ordinary[: n // 2] = False
    direction[ordinary] = (target - origin)[ordinary]
    rays["origin"], rays["direction"], rays["t"] = origin, direction, np.float32(1e30)
    got = ctx.trace_rays_packet(rays)
    want = _oracle().trace_rays(scene.desc, rays)
    for field in ("found", "triangle"):
        assert (got[field] == want[field]).all(), field
    assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    assert (got["tclosest"].view(np.uint32) == want["tclosest"].view(np.uint32)).all()
    assert want["found"].sum() > n // 20
    shadowed = ctx.trace_rays_packet(rays, any_hit=True)
    assert (shadowed["found"] == want["found"]).all()


@pytest.mark.parametrize("name,w,h", [("bunny", 320, 180), ("import", 256, 144)])
def test_random_views(name, w, h, scenes, gpu_ctx):
    """Many camera orientations and positions (every direction component non-zero, so the binary32
    filter of the packet walk decides nearly everything): each frame equals the oracle's."""
    scene, ctx = scenes(name), gpu_ctx(name)
    rng = np.random.default_rng(2026)
    base = np.array(scene.camera, dtype=np.float64)
    for view in range(6):
        angles = [np.float32(a) for a in rng.uniform(-0.35, 0.35, 3)]
        rot = _oracle().camera_rotation(*angles)
        for ss in (False, True):
            frame = scene.frame(w, h, threads=4, supersampling=ss, rotation=rot, rng_seed=view)
            jitter = rng.uniform(-0.05, 0.05, 3) * max(1.0, np.abs(base).max())
            frame.cam_pos = (C.c_double * 3)(*(base + jitter))
            gpu_fb, st = ctx.render(frame)
            cpu_fb, cst = _oracle().render(scene.desc, frame, os_threads=4)
            _compare(gpu_fb, cpu_fb, "%s view %d ss=%d" % (name, view, ss))
            assert 0 < st.rays_shadow <= cst.rays_shadow          # (the oracle also traces the row the bitmap drops)


def test_deep_chain_tree(monkeypatch):
    """Triangles spaced geometrically along x make the midpoint builder peel one off per level: a
    tree about as deep as it is large.  The packet walk then has dozens of entries pending (more
    than its LDS levels: the rest spill to HBM), the per-lane walk a deep stack of its own."""
    n = 56
    xs = 2.2 ** np.arange(n) * 0.02
    verts = np.array([[x, -1.0, 4.0 + 0.01 * i, x + 0.6 * x + 0.01, -1.0, 4.0 + 0.01 * i, x + 0.3 * x, 1.0, 4.0 + 0.01 * i]
                      for i, x in enumerate(xs)])
    scene = ca.HostScene(vertices=verts)
    depth = scene.lib.cth_tree_depth(scene.handle)
    assert depth >= 40
    frame = scene.frame(256, 128, threads=4, supersampling=False)
    frame.cam_pos = (C.c_double * 3)(1.0, 0.0, -6.0)
    want, _ = _oracle().render(scene.desc, frame, os_threads=2)
    assert (want != 0x333333).any()
    for env in ({}, {"CTGPU_PACKET_LDS_LEVELS": "2"}, {"CTGPU_PACKET": "0", "CTGPU_LDS_DEPTH": "2"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = ca.GpuContext(0)
        for k in env:
            monkeypatch.delenv(k)
        ctx.upload(scene.desc)
        try:
            for ss in (False, True):
                f = scene.frame(256, 128, threads=4, supersampling=ss, rng_seed=4)
                f.cam_pos = (C.c_double * 3)(1.0, 0.0, -6.0)
                got, _ = ctx.render(f)
                ref, _ = _oracle().render(scene.desc, f, os_threads=2)
                _compare(got, ref, "chain tree %s ss=%d" % (env, ss))
        finally:
            ctx.close()


def test_degenerate_inputs(tmp_path):
    """Zero-area triangles, a directional light with the zero vector as direction (every shadow
    ray is all-NaN arithmetic), an axis-parallel light direction (a zero component in every shadow
    ray: exact level only), a point light sitting on a surface: whatever the reference does with
    these, every GPU path does the same."""
    tris = [([-2, -1, 1], [2, -1, 1], [2, -1, 6], (255, 255, 255), 100), ([-2, -1, 1], [2, -1, 6], [-2, -1, 6], (240, 240, 240), 100)]
    for i in range(5):
        for j in range(4):
            x, z = -1.6 + 0.7 * i, 2.0 + 0.9 * j
            tris.append(([x, -1, z], [x + 0.4, -1, z], [x + 0.2, 0.2 + 0.1 * j, z], (200, 60 + 30 * i, 60), 20))
    tris.append(([0, 0, 3], [0, 0, 3], [0, 0, 3], (1, 2, 3), 5))                     # a point
    tris.append(([-1, 0.5, 4], [1, 0.5, 4], [0, 0.5, 4], (3, 2, 1), 5))              # a segment
    cases = {
        "zero_direction": [{"type": "ambient", "intensity": 0.3}, {"type": "directional", "intensity": 0.5, "direction": [0, 0, 0]}],
        "axis_direction": [{"type": "ambient", "intensity": 0.2}, {"type": "directional", "intensity": 0.6, "direction": [0, 1, 0]},
                           {"type": "directional", "intensity": 0.2, "direction": [1, 0, 0]}],
        "light_on_floor": [{"type": "ambient", "intensity": 0.2}, {"type": "point", "intensity": 0.7, "position": [0.25, -1, 3.5]}],
    }
    for name, lights in cases.items():
        path = _write_scene(tmp_path, name + ".json", tris, lights, camera=(0, 0.4, -3))
        for packet in ("1", "0"):
            with _knob("CTGPU_PACKET", packet):
                for ss in (False, True):
                    gpu_fb, cpu_fb, st, cst = _render_both(path, 192, 144, threads=4, supersampling=ss, rng_seed=3)
                    _compare(gpu_fb, cpu_fb, "%s packet=%s ss=%d" % (name, packet, ss))


@pytest.mark.parametrize("spheres,reflect_fix", [(True, False), (False, True), (True, True)])
def test_spheres_and_reflection_fix(spheres, reflect_fix, scenes, gpu_ctx, tmp_path):
    """SURVEY.md 8f rank 4.  CTGPU_F_SPHERES: the scene's sphere objects rendered through their own tree
    (IntersectRaySphere raythread.cpp:158-174, acceptance rule of raythread.cpp:230-246, sphere normal :331-334);
    CTGPU_F_REFLECT_FIX: reflected rays start at ray.t = 1e30f instead of the 0 of raythread.cpp:373.  Both off by
    default (the reference's pictures); on, the GPU's general path against the oracle's restatement."""
    scene, ctx = scenes("cube"), gpu_ctx("cube")               # 37 triangles, 109 spheres, reflective materials
    for ss, w, h in ((False, 640, 480), (True, 320, 240)):
        frame = scene.frame(w, h, supersampling=ss, rng_seed=2, spheres=spheres, reflect_fix=reflect_fix)
        plain = scene.frame(w, h, supersampling=ss, rng_seed=2)
        got, st = ctx.render(frame)
        want, cst = _oracle().render(scene.desc, frame, os_threads=8)
        _compare(got, want, "cube spheres=%d reflect_fix=%d ss=%d" % (spheres, reflect_fix, ss))
        assert ((got == 0) == (want == 0)).all()
        default, _ = ctx.render(plain)
        assert (default != got).sum() > 200                    # the flags really change the picture
        assert st.rays_primary == cst.rays_primary - (2 * (h // 2)) * (16 if ss else 1)
    # a scene with spheres in front of everything, lit by a point light (shadow rays stop at tMax = 1 for spheres)
    import json
    objs = [{"type": "sphere", "center": [0, -1, 3], "radius": 1, "color": [255, 0, 0], "specular": 500, "reflection": 0.2},
            {"type": "sphere", "center": [2, 0, 4], "radius": 1, "color": [0, 0, 255], "specular": 500, "reflection": 0.3},
            {"type": "sphere", "center": [-2, 0, 4], "radius": 1, "color": [0, 255, 0], "specular": 10, "reflection": 0.4},
            {"type": "sphere", "center": [0, -5001, 0], "radius": 5000, "color": [255, 255, 0], "specular": 1000, "reflection": 0.5},
            {"type": "sphere", "center": [0.5, 0.8, 2.2], "radius": 0.3, "color": [200, 200, 200], "specular": -1, "reflection": 0},
            {"type": "triangle", "p1": [-3, -1, 6], "p2": [3, -1, 6], "p3": [0, 3, 6], "color": [255, 0, 255], "specular": 50,
             "reflection": 0.1}]
    lights = [{"type": "ambient", "intensity": 0.2}, {"type": "point", "intensity": 0.6, "position": [2, 1, 0]},
              {"type": "directional", "intensity": 0.2, "direction": [1, 4, 4]}]
    path = tmp_path / "spheres.json"
    path.write_text(json.dumps({"objects": objs, "lights": lights, "camera": {"position": [0, 0, -1]},
                                "settings": {"numberOfThreads": 4, "subsampling": False, "wireframe": False, "supersampling": True}}))
    own = ca.HostScene(str(path))
    assert own.desc.n_spheres == 5
    c = ca.GpuContext(0)
    c.upload(own.desc)
    try:
        for ss in (False, True):
            frame = own.frame(400, 300, supersampling=ss, rng_seed=5, spheres=spheres, reflect_fix=reflect_fix)
            got, _ = c.render(frame)
            want, _ = _oracle().render(own.desc, frame, os_threads=4)
            _compare(got, want, "five spheres spheres=%d reflect_fix=%d ss=%d" % (spheres, reflect_fix, ss))
            if spheres:
                assert ((got & 0xff) > 150).sum() > 2000           # the red one is in the picture
    finally:
        c.close()


def _same_tree(scene):
    nodes, indexes, ms = ca.build_bvh_on_device(scene.vertices())
    want_nodes = scene.nodes()
    want_nodes = want_nodes.reshape(-1, 64)[:, :60]            # bvh_node_t without its 4 bytes of padding
    assert len(nodes) == scene.desc.n_nodes
    assert (indexes == scene.indexes()).all()
    assert (nodes[:, :60] == want_nodes).all()
    return ms


@pytest.mark.parametrize("name", ["cube", "import", "bunny", "dragon"])
def test_device_bvh_build_is_the_reference_tree(name, scenes):
    """ctgpu_build_bvh (ct_build.hip): BuildBVH (bvh.cpp:16-121) level by level on the GPU.  Byte for byte the arrays
    the host builder returns -- itself pinned to the reference's node / leaf counts and depth (test_oracle_kat) --
    boxes, child links, leaf ranges and the index permutation the reference's swap loop leaves."""
    ms = _same_tree(scenes(name))
    print("%s: device build %.2f ms" % (name, ms))


def test_device_bvh_build_on_awkward_inputs():
    """Coincident centroids (a side stays empty: the node remains a leaf but its indexes were permuted all the
    same), triangles in a plane (zero extent on an axis), -0.0 against +0.0 corners (the a<b?a:b fold keeps the
    LATER of equals), one- and two-triangle trees, a deep chain, random soups of many sizes."""
    rng = np.random.default_rng(5)
    cases = {
        "one": rng.random((1, 9)), "two": rng.random((2, 9)), "three": rng.random((3, 9)),
        "same": np.tile(rng.random((1, 9)), (40, 1)),
        "same centroid, different boxes": np.array([[-k, -k, 0, k, -k, 0, 0, 2 * k, 0.0] for k in range(1, 30)]),
        "flat": np.c_[rng.random((500, 2)), np.zeros(500), rng.random((500, 2)), np.zeros(500), rng.random((500, 2)), np.zeros(500)],
        "chain": np.array([[x, -1.0, 4.0, 1.6 * x + 0.01, -1.0, 4.0, 1.3 * x, 1.0, 4.0] for x in 2.2 ** np.arange(56) * 0.02]),
    }
    zeros = rng.random((300, 9)) - 0.5
    zeros[rng.random((300, 9)) < 0.3] = 0.0
    zeros[rng.random((300, 9)) < 0.15] = -0.0                      # sign of zero: which corner comes later decides
    cases["signed zeros"] = zeros
    for n in (7, 64, 65, 1000, 4097, 50_000):
        cases["soup %d" % n] = rng.normal(size=(n, 9)) * np.array([3, 1, 0.2] * 3) + np.repeat(rng.normal(size=(n, 3)) * 4, 3, axis=1).reshape(n, 9)
    for name, verts in cases.items():
        scene = ca.HostScene(vertices=np.ascontiguousarray(verts, dtype=np.float64))
        nodes, indexes, _ = ca.build_bvh_on_device(scene.vertices())
        assert len(nodes) == scene.desc.n_nodes, name
        assert (indexes == scene.indexes()).all(), name
        assert (nodes[:, :60] == scene.nodes().reshape(-1, 64)[:, :60]).all(), name


def test_random_scenes_fuzz():
    """tools/fuzz_parity.py for a few seconds (seeds 50001 ..): random triangle soups of many sizes and coordinate
    scales, cameras anywhere, random lights (axis-parallel ones included) and materials (reflective ones take the
    general path), frames and single rays through all three kernels -- no difference from the oracle.  The long run
    of the same tool (2,263 scenes, 27.8 M rays, 0 differences) is in profiles/r02_fuzz.log."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "8", "50001"], capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    assert "no difference beyond" in proc.stdout and proc.stdout.count("\nseed ") >= 10
