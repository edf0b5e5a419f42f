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
    monkeypatch.setenv("CTGPU_PACKET", "1")                   # (the hand-over belongs to the shared walk)
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


def test_spheres_stay_on_the_fast_path(tmp_path, monkeypatch):
    """CTGPU_F_SPHERES frames of a scene without reflective materials: the triangles on the packet walk, the spheres
    in their own passes after it (ct_sphere_pass) -- against the oracle's restatement and against the general path
    (which renders the same frame in one kernel).  Spheres in front of, behind and between triangles; shadows of
    spheres on triangles and of triangles on spheres; a point light (tMax = 1) and a directional one."""
    import json
    from oracle import oracle
    rng = np.random.default_rng(12)
    objs = [{"type": "sphere", "center": [0, -1, 3], "radius": 1, "color": [255, 0, 0], "specular": 500, "reflection": 0},
            {"type": "sphere", "center": [2, 0, 4], "radius": 1, "color": [0, 0, 255], "specular": 500, "reflection": 0},
            {"type": "sphere", "center": [-2, 0, 4], "radius": 1, "color": [0, 255, 0], "specular": 10, "reflection": 0},
            {"type": "sphere", "center": [0, -5001, 0], "radius": 5000, "color": [255, 255, 0], "specular": 1000, "reflection": 0},
            {"type": "sphere", "center": [0.5, 0.8, 2.2], "radius": 0.3, "color": [200, 200, 200], "specular": -1, "reflection": 0},
            {"type": "sphere", "center": [0.2, 1.5, 8.0], "radius": 1.2, "color": [90, 200, 200], "specular": 40, "reflection": 0}]
    for _ in range(60):                                     # a cloud of small triangles among the spheres
        c = rng.uniform([-3, -1, 2], [3, 2.5, 7])
        p = [list(c + rng.normal(size=3) * 0.35) for _ in range(3)]
        objs.append({"type": "triangle", "p1": p[0], "p2": p[1], "p3": p[2], "color": [int(x) for x in rng.integers(60, 255, 3)],
                     "specular": int(rng.choice([-1, 20, 300])), "reflection": 0})
    objs.append({"type": "triangle", "p1": [-4, -1, 6.5], "p2": [4, -1, 6.5], "p3": [0, 4, 6.5], "color": [255, 0, 255], "specular": 50, "reflection": 0})
    lights = [{"type": "ambient", "intensity": 0.2}, {"type": "point", "intensity": 0.6, "position": [2, 1, 0]},
              {"type": "directional", "intensity": 0.2, "direction": [1, 4, 4]}]
    path = tmp_path / "spheres_and_triangles.json"
    path.write_text(json.dumps({"objects": objs, "lights": lights, "camera": {"position": [0, 0, -1]},
                                "settings": {"numberOfThreads": 4, "subsampling": False, "wireframe": False, "supersampling": True}}))
    scene = ca.HostScene(str(path))
    assert scene.desc.n_spheres == 6 and scene.desc.n_triangles == 61
    monkeypatch.setenv("CTGPU_PACKET", "1")
    monkeypatch.delenv("CTGPU_FORCE_MEGAKERNEL", raising=False)
    fast = ca.GpuContext(0)
    fast.upload(scene.desc)
    monkeypatch.setenv("CTGPU_FORCE_MEGAKERNEL", "1")
    general = ca.GpuContext(0)
    general.upload(scene.desc)
    monkeypatch.delenv("CTGPU_FORCE_MEGAKERNEL")
    try:
        for ss, w, h in ((False, 640, 480), (True, 400, 300)):
            frame = scene.frame(w, h, supersampling=ss, rng_seed=5, spheres=True)
            fast.read_walk_work()
            got, st = fast.render(frame)
            assert fast.read_walk_work().packets[0] > 0                     # the packet walk traced the triangles
            assert st.walk_kind == ca.bindings.WALK_PACKET and st.bounce_kind == ca.bindings.BOUNCE_NONE
            want, _ = oracle.render(scene.desc, frame, os_threads=4)
            other, _ = general.render(frame)
            diff = int((got != want).sum())
            print("spheres on the fast path ss=%d: %d pixels differ from the oracle, %d from the general path" % (ss, diff, int((got != other).sum())))
            shifts = np.array([0, 8, 16])
            assert np.abs(((got[..., None] >> shifts) & 255).astype(int) - ((want[..., None] >> shifts) & 255).astype(int)).max() <= 1
            assert (got == other).all()
            without, _ = fast.render(scene.frame(w, h, supersampling=ss, rng_seed=5))
            assert (without != got).sum() > 2000                            # the flag really draws spheres
    finally:
        fast.close()
        general.close()


@pytest.mark.parametrize("spheres,reflect_fix", [(False, False), (True, False), (False, True), (True, True)])
def test_reflection_bounces_on_the_fast_path(spheres, reflect_fix, assets, monkeypatch):
    """Reflective materials (scene_file_cube.json, scene_file_1.json): TraceRay's recursion (raythread.cpp:353-386,
    the t = 0 start of reflected rays included) as bounce levels on the wavefront pipeline -- level 0 through the frame
    walks, levels 1..10 through the packet walk's ray-list kernels -- against the oracle (<= 1 LSB) and against the
    general path's one kernel (identical pixels, identical ray counts)."""
    import os
    from oracle import oracle
    monkeypatch.delenv("CTGPU_FORCE_MEGAKERNEL", raising=False)
    monkeypatch.delenv("CTGPU_BOUNCE", raising=False)
    shifts = np.array([0, 8, 16])
    for name in ("scene_file_cube.json", "scene_file_1.json"):
        scene = ca.HostScene(os.path.join(assets, name))
        monkeypatch.setenv("CTGPU_PACKET", "1")
        monkeypatch.setenv("CTGPU_BOUNCE", "1")
        fast = ca.GpuContext(0)
        fast.upload(scene.desc, build_on_device=True)
        monkeypatch.setenv("CTGPU_BOUNCE", "0")
        general = ca.GpuContext(0)
        general.upload(scene.desc)
        monkeypatch.delenv("CTGPU_BOUNCE")
        try:
            for ss, w, h in ((False, 640, 480), (True, 320, 240)):
                frame = scene.frame(w, h, supersampling=ss, rng_seed=4, spheres=spheres, reflect_fix=reflect_fix)
                fast.read_walk_work()
                got, st = fast.render(frame)
                work = fast.read_walk_work()
                other, so = general.render(frame)
                assert (st.walk_kind, st.bounce_kind) == (ca.bindings.WALK_PACKET, ca.bindings.BOUNCE_LEVELS)
                assert (so.walk_kind, so.bounce_kind) == (ca.bindings.WALK_GENERAL, ca.bindings.BOUNCE_GENERAL)
                want, _ = oracle.render(scene.desc, frame, os_threads=8)
                diff = np.abs(((got[..., None] >> shifts) & 255).astype(int) - ((want[..., None] >> shifts) & 255).astype(int)).max()
                print("%s spheres=%d fix=%d ss=%d: fast path %.3f ms, general path %.3f ms; %d px differ from the oracle, %d from the general path; "
                      "%d reflected rays" % (name, spheres, reflect_fix, ss, st.kernel_ms, so.kernel_ms, int((got != want).sum()),
                                             int((got != other).sum()), st.rays_reflect))
                assert diff <= 1 and ((got == 0) == (want == 0)).all()
                assert (got == other).all()
                assert work.packets[0] > 0                                  # the frame's primary rays went through the packet walk
                assert (st.rays_primary, st.rays_reflect) == (so.rays_primary, so.rays_reflect) and st.rays_reflect > 0
                assert st.rays_shadow + st.rays_shadow_elided == so.rays_shadow + so.rays_shadow_elided
        finally:
            fast.close()
            general.close()


def test_reflective_frames_take_the_faster_path(assets, monkeypatch):
    """Without CTGPU_BOUNCE a context renders two reflective frames of a kind on the bounce pipeline, two on the
    general path, and keeps whichever took less kernel time; the pixels never change."""
    import os
    monkeypatch.delenv("CTGPU_BOUNCE", raising=False)
    monkeypatch.delenv("CTGPU_FORCE_MEGAKERNEL", raising=False)
    scene = ca.HostScene(os.path.join(assets, "scene_file_cube.json"))
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    try:
        for ss, w, h in ((True, 640, 480), (False, 320, 240)):
            frame = scene.frame(w, h, supersampling=ss, rng_seed=1)
            frames = [ctx.render(frame)[0] for _ in range(7)]
            assert all((f == frames[0]).all() for f in frames[1:])
            general, pipeline_ms, general_ms = ctx.bounce_choice(ss)
            print("cube %dx%d ss=%d: bounce levels %.3f ms, general path %.3f ms -> %s" % (w, h, ss, pipeline_ms, general_ms,
                  "general path" if general else "bounce levels"))
            assert pipeline_ms > 0 and general_ms > 0
            assert general == (general_ms < pipeline_ms) or abs(general_ms - pipeline_ms) < 0.15 * pipeline_ms      # (the figures move on after the decision)
    finally:
        ctx.close()
