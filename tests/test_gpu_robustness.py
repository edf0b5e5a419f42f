"""-m gpu: what the library does when something goes wrong around a frame (round-2 advisor findings) -- a rank
whose part of a sharded frame fails still takes part in the gather, a missing RCCL leaves the one-process
multi-device frame working, a framebuffer's page-lock ends with its frame, a context with a frame in flight
refuses a new scene."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_failing_rank_still_gathers_and_reports(scenes, gpu_ctx):
    """rank 0 without a framebuffer is an error only rank 0 can see: it must still queue its ncclGather (the other
    ranks would wait in theirs for ever) and then report; the context stays usable."""
    torch = pytest.importorskip("torch")
    scene = scenes("bunny")
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    try:
        ctx.comm_init(ca.GpuContext.comm_unique_id(), 0, 1)
        frame = scene.frame(320, 180, threads=4, supersampling=False)
        stream = torch.cuda.Stream()
        with pytest.raises(RuntimeError, match="rank 0 needs a framebuffer"):
            ctx.render_sharded(frame, 8, 0, stream.cuda_stream)
        stream.synchronize()                                            # the gather it queued has run
        bad = scene.frame(320, 180, threads=4)
        bad.num_threads = 0                                             # same verdict on every rank: nothing is queued
        with pytest.raises(RuntimeError, match="num_threads"):
            ctx.render_sharded(bad, 8, 0, stream.cuda_stream)
        fb = torch.zeros(frame.height * frame.width, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.kernel_timing(True)
        for _ in range(3):
            ctx.render_sharded(frame, 8, fb.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        phases = ctx.read_phase_times()
        ctx.kernel_timing(False)
        ctx.read_counters()
        want, _ = gpu_ctx("bunny").render(frame)
        assert (fb.cpu().numpy().view(np.uint32).reshape(frame.height, frame.width) == want).all()
        assert phases.frames == 3 and phases.trace_ms > 0 and phases.gather_ms >= 0 and phases.assemble_ms > 0
        assert ctx.read_phase_times().frames == 0                       # read = reset
    finally:
        ctx.close()


def test_without_rccl_the_multi_device_frame_still_renders():
    """CTGPU_RCCL_LIBRARY pointing nowhere = no RCCL in this process (own process: the binding happens once):
    the comm entry points fail with CTGPU_ERR_COMM, ctgpu_render_multi hands tiles over with device copies."""
    code = textwrap.dedent("""
        import os, sys
        import numpy as np
        sys.path.insert(0, %r)
        import cobbletrace_amd as ca
        scene = ca.HostScene(os.path.join(%r, "assets", "scene_import.json"))
        try:
            ca.GpuContext.comm_unique_id()
            raise SystemExit("comm_unique_id worked without RCCL")
        except RuntimeError as e:
            assert "CTGPU_RCCL_LIBRARY" in str(e) and "error -6" in str(e), str(e)
        ctxs = [ca.GpuContext(0) for _ in range(3)]
        for c in ctxs:
            c.upload(scene.desc)
        frame = scene.frame(400, 300, threads=10, rng_seed=4)
        one, _ = ctxs[0].render(frame)
        many, st = ca.GpuContext.render_multi(ctxs, frame)
        assert (one == many).all() and st.rays > 0
        print("ok")
    """ % (ROOT, ROOT))
    env = dict(os.environ, CTGPU_RCCL_LIBRARY="/nonexistent/librccl.so")
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0 and "ok" in proc.stdout, proc.stdout + proc.stderr


def test_framebuffer_pin_ends_with_the_frame(scenes):
    """ctgpu_render_begin page-locks the caller's framebuffer for the frame only: arrays that come and go at one
    address (numpy reuses freed blocks) never meet a stale registration; ctgpu_keep_framebuffer_pinned keeps it
    for hosts with one long-lived framebuffer."""
    scene = scenes("import")
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    try:
        frame = scene.frame(640, 480, threads=8, supersampling=False)
        want, _ = ctx.render(frame)
        seen = set()
        for _ in range(6):
            fb = ctx.render_begin(frame)                    # a fresh array per frame, kept alive by the wrapper
            seen.add(fb.ctypes.data)
            with pytest.raises(RuntimeError, match="already in flight"):
                ctx.render_begin(frame)
            with pytest.raises(RuntimeError, match="frame is in flight"):
                ctx.upload(scene.desc)
            done, st = ctx.render_poll(wait=True)
            assert done and (fb == want).all()
            del fb
        ctx.keep_framebuffer_pinned(True)
        fb = np.zeros((480, 640), dtype=np.uint32)
        for _ in range(3):
            ctx.render_begin(frame, fb)
            assert ctx.render_poll(wait=True)[0] and (fb == want).all()
        ctx.keep_framebuffer_pinned(False)
        ctx.render_begin(frame, fb)
        assert ctx.render_poll(wait=True)[0] and (fb == want).all()
        with pytest.raises(RuntimeError, match="no frame is in flight"):
            ctx.render_poll()
    finally:
        ctx.close()


def test_pageable_framebuffers_in_the_c_heap():
    """ctgpu_render into arrays that live in the C heap and come and go while the heap's top is given back to the system
    and grown again: the frame is staged in page-locked memory of the context's own and copied over by the CPU, so no
    copy queued on a stream ever targets such memory (the runtime's own pinning of it went stale in exactly this
    pattern: "Memory access fault by GPU" late in long test sessions).  ctgpu_download_frame, whose copy IS queued on
    the caller's stream, insists on page-locked memory.  In a process of its own: it changes malloc's thresholds."""
    code = textwrap.dedent("""
        import ctypes, os, sys
        import numpy as np
        libc = ctypes.CDLL(None)
        assert libc.mallopt(-3, 1 << 30) == 1            # M_MMAP_THRESHOLD: arrays of every size from the brk heap
        assert libc.mallopt(-1, 64 << 10) == 1           # M_TRIM_THRESHOLD: give the top back as soon as 64 KB are free
        sys.path.insert(0, %r)
        import cobbletrace_amd as ca
        scene = ca.HostScene(os.path.join(%r, "assets", "scene_import.json"))
        ctx = ca.GpuContext(0)
        ctx.upload(scene.desc, build_on_device=True)
        frames = [scene.frame(w, h, threads=8, supersampling=ss, rng_seed=3) for (w, h, ss) in ((640, 480, False), (960, 540, False), (320, 240, True))]
        want = [ctx.render(f)[0].copy() for f in frames]
        rng = np.random.default_rng(1)
        for k in range(60):
            ballast = np.ones(int(rng.integers(1, 24)) << 18, dtype=np.uint32)         # 1 .. 24 MB below the framebuffer
            i = k %% len(frames)
            fb = np.zeros((frames[i].height, frames[i].width), dtype=np.uint32)
            got, _ = ctx.render(frames[i], fb)
            assert got is fb and (fb == want[i]).all(), k
            if k %% 3 == 0:
                del ballast                                                        # the top of the heap goes back ...
            del fb, got                                                            # ... and comes again next round
        # A registration that outlives its memory (advisor, round 3): a context that keeps its page-lock between frames
        # draws into a C-heap array, the array is freed, and a LARGER array comes to lie on the same addresses.  The
        # runtime would still call its first byte "host memory"; the library knows the range as one of its own
        # registrations and stages -- in the context that made it, in another context, through begin / poll and render.
        keeper = ca.GpuContext(0)
        keeper.share_scene(ctx)
        keeper.keep_framebuffer_pinned(True)
        small = np.zeros((frames[2].height, frames[2].width), dtype=np.uint32)
        at = small.ctypes.data
        keeper.render_begin(frames[2], small)
        assert keeper.render_poll(wait=True)[0] and (small == want[2]).all()
        del small                                                                  # freed; the page-lock stays (keep_pin)
        big = np.zeros((frames[1].height, frames[1].width), dtype=np.uint32)      # a larger array from the same heap
        reused = big.ctypes.data <= at < big.ctypes.data + big.nbytes
        for renderer in (ctx, keeper):
            got, _ = renderer.render(frames[1], big)
            assert (big == want[1]).all()
            big[:] = 0
        keeper.render_begin(frames[1], big)                                        # another range: the stale pin is dropped, big registered
        assert keeper.render_poll(wait=True)[0] and (big == want[1]).all()
        keeper.keep_framebuffer_pinned(False)
        keeper.close()
        print("stale registration: the larger array", "covers" if reused else "does not cover", "the freed one's addresses")
        try:
            ctx.download_frame(frames[0], 1, np.zeros((480, 640), dtype=np.uint32).ctypes.data)
        except RuntimeError as e:
            assert "page-locked" in str(e), e
        else:
            raise AssertionError("ctgpu_download_frame took pageable memory")
        ctx.close()
        print("ok")
    """) % (ROOT, ROOT)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and proc.stdout.strip().endswith("ok"), proc.stdout[-2000:] + proc.stderr[-4000:]


def test_contexts_report_and_take_their_configuration(scenes, monkeypatch):
    """ctgpu_get_config: a context made by ctgpu_create runs with what the environment said at that moment;
    ctgpu_create_with_config takes the caller's struct instead (and refuses values out of range); the knobs do what the
    header says -- here: no cull nodes with big_leaf_min = 0, the host's packing with pack_on_host -- and no pixel moves."""
    for name in list(os.environ):                # (the suite is also run with knobs in the environment: this test owns them all)
        if name.startswith("CTGPU_") and name not in ("CTGPU_LIBRARY", "CTGPU_RCCL_LIBRARY"):
            monkeypatch.delenv(name)
    monkeypatch.setenv("CTGPU_CHUNK", "256")
    monkeypatch.setenv("CTGPU_ORDER", "0")
    ctx = ca.GpuContext(0)
    monkeypatch.delenv("CTGPU_CHUNK")
    monkeypatch.delenv("CTGPU_ORDER")
    scene = scenes("import")
    try:
        got = ctx.get_config()
        assert (got.chunk, got.order) == (256, 0)
        want = ca.default_config()
        want.chunk, want.order = 256, 0
        assert got.as_dict() == want.as_dict()
        ctx.upload(scene.desc)
        frame = scene.frame(320, 240, rng_seed=2)
        reference, _ = ctx.render(frame)
        assert len(ctx.scene_array(6)) > 0
        custom = ca.default_config()
        custom.big_leaf_min, custom.pack_on_host, custom.packet_walk, custom.keep_shadow_rays = 0, 1, 1, 1
        other = ca.GpuContext(0, config=custom)
        try:
            assert other.get_config().as_dict() == custom.as_dict()
            other.upload(scene.desc)
            assert len(other.scene_array(6)) == 0
            fb, _ = other.render(frame)
            assert (fb == reference).all()
        finally:
            other.close()
        bad = ca.default_config()
        bad.order_parts[1] = 3
        with pytest.raises(RuntimeError, match="out of range"):
            ca.GpuContext(0, config=bad)
    finally:
        ctx.close()
