"""-m gpu: heavy chunks first (ctgpu_order_info).  Frames with few packets per wave hand the chunks of ray ids the frame
before found heavy out first, the heaviest cut into parts; none of it may change a pixel or a ray count -- whatever the
thresholds, when the lists overflow, and when the history belongs to another frame."""
import ctypes as C

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu


def _context(monkeypatch, scene, **knobs):
    base = {"CTGPU_PACKET": 1, "CTGPU_PACKET_BUDGET": None, "CTGPU_ORDER": None, "CTGPU_ORDER_HEAVY": None,
            "CTGPU_ORDER_PARTS": None, "CTGPU_ORDER_FLOOR": None, "CTGPU_ORDER_SPLIT_BELOW": None, "CTGPU_CHUNK": None,
            "CTGPU_CHUNK_SHADOW": None}
    base.update(knobs)
    for key, value in base.items():
        if value is None:
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, str(value))
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=True)
    return ctx


def _counts(st):
    return (st.rays_primary, st.rays_shadow, st.rays_shadow_elided, st.rays_reflect)


@pytest.mark.parametrize("name,w,h,ss", [("bunny", 1920, 1080, False), ("dragon", 1920, 1080, False), ("import", 480, 270, True)])
def test_order_keeps_every_pixel(name, w, h, ss, scenes, monkeypatch):
    scene = scenes(name)
    plain = _context(monkeypatch, scene, CTGPU_ORDER=0)
    ordered = _context(monkeypatch, scene)
    try:
        frame = scene.frame(w, h, supersampling=ss, rng_seed=4)
        want, wst = plain.render(frame)
        assert plain.order_info() == ([0] * 8, False) and not wst.ordered
        for k in range(4):
            got, st = ordered.render(frame)
            assert st.ordered == 1 and st.walk_kind == ca.bindings.WALK_PACKET        # (ctgpu_stats says which code ran)
            assert (got == want).all(), "frame %d" % k
            assert _counts(st) == _counts(wst)
            entries, split = ordered.order_info()
            assert split
            if k == 0:
                assert entries == [0] * 8                    # nothing is known before the first frame
        # both walks had heavy chunks to put first (the shadow tree's packets of this small scene stay below the floor of 64
        # fetches a heavy chunk needs: tests/test_gpu_shadow_tree.py has the ordered shadow-tree frame with lists)
        assert sum(entries[:4]) > 0 and (sum(entries[4:]) > 0 or st.shadow_tree), entries
        print("%s %dx%d ss=%d: list entries %s" % (name, w, h, ss, entries))
    finally:
        plain.close()
        ordered.close()


def test_order_with_everything_heavy_and_full_lists(scenes, monkeypatch):
    """Thresholds far below the mean and sixteen parts per chunk: every chunk that did anything wants sixteen list slots,
    the lists (one slot per chunk) overflow, the rest stays in the usual run -- every ray still traced exactly once."""
    scene = scenes("bunny")
    plain = _context(monkeypatch, scene, CTGPU_ORDER=0)
    eager = _context(monkeypatch, scene, CTGPU_ORDER_HEAVY="0.4,0.3,0.2,0.1", CTGPU_ORDER_PARTS="16,16,8,2", CTGPU_ORDER_FLOOR=1)
    try:
        for w, h, ss in ((960, 540, False), (320, 180, True)):
            frame = scene.frame(w, h, supersampling=ss, rng_seed=9)
            want, wst = plain.render(frame)
            for k in range(3):
                got, st = eager.render(frame)
                assert (got == want).all() and _counts(st) == _counts(wst), (w, h, ss, k)
            entries, split = eager.order_info()
            n_chunks = (wst.rays_primary + 63) // 64
            assert split and entries[0] > n_chunks // 2, (entries, n_chunks)     # (asked for far more than the list's room)
    finally:
        plain.close()
        eager.close()


def test_order_history_belongs_to_one_frame_geometry(scenes, monkeypatch):
    """Costs noted for one frame say nothing about another size, sampling mode or partition: those start afresh; a moved
    camera keeps the (now partly wrong) history, which may only cost time."""
    scene = scenes("bunny")
    plain = _context(monkeypatch, scene, CTGPU_ORDER=0)
    ordered = _context(monkeypatch, scene)
    second = _context(monkeypatch, scene)
    try:
        a = scene.frame(1280, 720, supersampling=False, rng_seed=1)
        b = scene.frame(640, 360, supersampling=False, rng_seed=1)
        c = scene.frame(400, 224, supersampling=True, rng_seed=1)
        moved = scene.frame(1280, 720, supersampling=False, rng_seed=1)
        moved.cam_pos = (C.c_double * 3)(a.cam_pos[0] + 0.8, a.cam_pos[1] - 0.3, a.cam_pos[2] + 0.5)
        for frame in (a, a, b, a, c, c, moved, a, moved, moved, b, b):
            want, wst = plain.render(frame)
            got, st = ordered.render(frame)
            assert (got == want).all() and _counts(st) == _counts(wst)
        # the same frames as row-tile partitions over two contexts of one device (each keeps the history of its own part)
        for frame in (a, a, moved, a):
            want, _ = plain.render(frame)
            got, _ = ca.GpuContext.render_multi([ordered, second], frame)
            assert (got == want).all()
            got, _ = ordered.render(frame)
            assert (got == want).all()
    finally:
        plain.close()
        ordered.close()
        second.close()


def test_order_is_for_frames_with_few_packets_per_wave(scenes, monkeypatch):
    scene = scenes("bunny")
    ctx = _context(monkeypatch, scene)
    never = _context(monkeypatch, scene, CTGPU_ORDER_SPLIT_BELOW=0)
    always = _context(monkeypatch, scene, CTGPU_ORDER=2)
    try:
        big = scene.frame(1920, 1080, supersampling=True, rng_seed=2)        # 142 k chunks for 7 k waves: left alone
        want, wst = ctx.render(big)
        ctx.render(big)
        assert ctx.order_info() == ([0] * 8, False)
        for k in range(3):                                                    # (CTGPU_ORDER=2: heavy chunks first, whole)
            got, st = always.render(big)
            assert (got == want).all() and _counts(st) == _counts(wst)
        entries, split = always.order_info()
        assert not split and sum(entries) > 0
        small = scene.frame(1920, 1080, supersampling=False, rng_seed=2)
        never.render(small)
        never.render(small)
        assert never.order_info() == ([0] * 8, False)
        ctx.render(small)
        ctx.render(small)
        assert ctx.order_info()[1] and sum(ctx.order_info()[0]) > 0
        ctx.render(big)                                                       # (the report is about the latest frame)
        assert ctx.order_info() == ([0] * 8, False)
    finally:
        ctx.close()
        never.close()
        always.close()
