"""-m gpu: which walk traces a frame -- the shared (packet) walk or the per-lane one -- is chosen by measurement, and
the experimental hand-over of incoherent packets to the per-lane kernel; neither may change a pixel or a ray count."""
import ctypes as C

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu


def _sheet(side, seed=3):
    """tools/scale_check.py's scene at a size that takes seconds: a bumpy sheet seen at a grazing angle, many
    sub-pixel triangles per pixel -- neighbouring rays share little of their walks."""
    rng = np.random.default_rng(seed)
    u, w = np.linspace(-8.0, 8.0, side + 1), np.linspace(4.0, 20.0, side + 1)
    X, Z = np.meshgrid(u, w, indexing="ij")
    Y = -1.5 + 0.4 * np.sin(X * 1.7) * np.cos(Z * 0.9) + rng.normal(size=X.shape) * (8.0 / side)
    P = np.stack([X, Y, Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    verts = np.concatenate([np.concatenate([a, b, c], axis=-1).reshape(-1, 9), np.concatenate([b, d, c], axis=-1).reshape(-1, 9)])
    return verts[rng.permutation(len(verts))]


@pytest.mark.parametrize("name,w,h", [("bunny", 960, 540), ("import", 640, 360)])
def test_handover_of_incoherent_packets_keeps_every_pixel(name, w, h, scenes, gpu_ctx, monkeypatch):
    scene = scenes(name)
    monkeypatch.setenv("CTGPU_PACKET_BUDGET", "16")
    monkeypatch.setenv("CTGPU_PACKET_LANES", "40")            # eager: most packets that live 16 visits hand over
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=True)
    try:
        for ss in (False, True):
            frame = scene.frame(w, h, supersampling=ss, rng_seed=6)
            want, wst = gpu_ctx(name).render(frame)
            ctx.read_walk_work()
            got, st = ctx.render(frame)
            work = ctx.read_walk_work()
            assert (got == want).all()
            assert (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (wst.rays_primary, wst.rays_shadow, wst.rays_shadow_elided)
            assert work.rays_handed_over[0] > 0 and work.rays_handed_over[1] > 0, list(work.rays_handed_over)
            assert work.rays_handed_over[0] <= st.rays_primary and work.rays_handed_over[1] <= st.rays_shadow
    finally:
        ctx.close()


def test_walk_is_chosen_by_measurement(scenes, monkeypatch):
    """On the sheet the per-lane walk wins (tools/scale_check.py: 1.9x on 8 M triangles): the context finds out from
    its own frame times and switches; the frames do not change.  On the bunny it stays with the shared walk, and with
    CTGPU_PACKET in the environment it never tries."""
    monkeypatch.delenv("CTGPU_PACKET", raising=False)
    monkeypatch.setenv("CTGPU_PROBE_ABOVE", "40")                  # (default 100: this sheet is a thirtieth of scale_check's)
    scene = ca.HostScene(vertices=_sheet(360))                     # 259,200 triangles
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=True)
    try:
        frame = scene.frame(1280, 720, threads=8, supersampling=False)
        frame.cam_pos = (C.c_double * 3)(0.0, 2.5, -4.0)
        ctx.read_walk_work()
        frames = [ctx.render(frame) for _ in range(6)]
        work = ctx.read_walk_work()
        print("sheet: fetches per packet while on the shared walk: %.1f" % ((work.quad_nodes[0] + work.quad_nodes[1] + work.triangle_records[0]
              + work.triangle_records[1]) / max(1, work.packets[0] + work.packets[1])))
        for fb, st in frames[1:]:
            assert (fb == frames[0][0]).all() and st.rays == frames[0][1].rays
        lane, packet_ms, lane_ms = ctx.walk_choice(False)
        print("sheet: per-lane chosen %s, shared %.3f ms, per-lane %.3f ms" % (lane, packet_ms, lane_ms))
        assert lane_ms > 0.0, "a scene with hundreds of fetches per packet must get its trial frame"
        # (the figures are the LAST frame on either walk; the decision was taken on the trial frame's, with a 10 % margin)
        assert (lane_ms < packet_ms) if lane else (lane_ms > 0.8 * packet_ms)
        if lane:                                                   # the frames after the trial ran per lane: no packets walked
            ctx.read_walk_work()
            ctx.render(frame)
            assert ctx.read_walk_work().packets[0] == 0
    finally:
        ctx.close()

    monkeypatch.delenv("CTGPU_PROBE_ABOVE")
    bunny = scenes("bunny")
    ctx = ca.GpuContext(0)
    ctx.upload(bunny.desc)
    try:
        frame = bunny.frame(960, 540, supersampling=True)
        for _ in range(6):
            ctx.render(frame)
        lane, packet_ms, lane_ms = ctx.walk_choice(True)
        assert not lane and packet_ms > 0.0
    finally:
        ctx.close()

    monkeypatch.setenv("CTGPU_PACKET", "1")
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    try:
        for _ in range(6):
            ctx.render(frame)
        assert ctx.walk_choice(True) == (False, 0.0, 0.0) or ctx.walk_choice(True)[2] == 0.0
    finally:
        ctx.close()
