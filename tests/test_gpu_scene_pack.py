"""-m gpu: the scene's walk layouts made on the device (ct_pack.hip), and the tree built there too (ct_build.hip),
against the host restatement of the same packing (CTGPU_PACK=host) -- every array byte for byte, every scalar, and
the frames that come out; one scene in HBM shared by several contexts."""
import ctypes as C
import time

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu
ARRAYS = {0: "WideNodes", 1: "QuadNodes", 2: "QuadNode boxes", 3: "unfolded QuadNodes", 4: "unfolded entries' reference nodes", 5: "TriRecords"}
SCALARS = ("n_triangles", "n_nodes", "tree_depth", "n_wide", "n_quads", "n_quads_flat", "quad_root", "quad_root_flat",
           "quad_pending", "root_ref", "slot_of_triangle0")


def _three_ways(desc, monkeypatch, build=True):
    """contexts holding the same scene packed on the host, packed on the device, and (build) built + packed there"""
    out = {}
    monkeypatch.setenv("CTGPU_PACK", "host")
    out["host"] = ca.GpuContext(0)
    out["host"].upload(desc)
    monkeypatch.delenv("CTGPU_PACK")
    out["device"] = ca.GpuContext(0)
    out["device"].upload(desc)
    if build:
        out["built"] = ca.GpuContext(0)
        out["built"].upload(desc, build_on_device=True)
    return out


def _same_scene(ctxs, what):
    ref = ctxs["host"]
    want = ref.scene_info()
    for name, ctx in ctxs.items():
        if name == "host":
            continue
        got = ctx.scene_info()
        for field in SCALARS:
            assert getattr(got, field) == getattr(want, field), (what, name, field, getattr(got, field), getattr(want, field))
        assert list(got.plane_bound) == list(want.plane_bound), (what, name)
        for which, label in ARRAYS.items():
            a, b = ref.scene_array(which), ctx.scene_array(which)
            assert a.shape == b.shape and (a == b).all(), (what, name, label, int((a != b).sum()))


@pytest.mark.parametrize("name", ["cube", "import", "bunny", "dragon"])
def test_device_packing_equals_the_host_packing(name, scenes, monkeypatch):
    scene = scenes(name)
    t0 = time.perf_counter()
    ctxs = _three_ways(scene.desc, monkeypatch)
    try:
        _same_scene(ctxs, name)
        nodes, indexes = ctxs["built"].download_tree()             # the reference's arrays, from the device build
        assert len(nodes) == scene.desc.n_nodes and (indexes == scene.indexes()).all()
        assert (nodes[:, :60] == scene.nodes().reshape(-1, 64)[:, :60]).all()
        with pytest.raises(RuntimeError, match="came from the caller"):
            ctxs["device"].download_tree()
        info = ctxs["built"].scene_info()
        print("%s: %d triangles, %d nodes -> %d + %d QuadNodes, depth %d, pending %d; BuildBVH %.2f ms, packing %.2f ms on the device; %.0f MB"
              % (name, info.n_triangles, info.n_nodes, info.n_quads, info.n_quads_flat, info.tree_depth, info.quad_pending,
                 info.build_ms, info.pack_ms, info.hbm_bytes / 1e6))
        w, h = (640, 360) if name != "cube" else (320, 240)
        for ss in (False, True):
            frame = scene.frame(w, h, supersampling=ss, rng_seed=5)
            frames = {k: c.render(frame)[0] for k, c in ctxs.items()}
            assert (frames["device"] == frames["host"]).all() and (frames["built"] == frames["host"]).all()
    finally:
        for c in ctxs.values():
            c.close()


def test_packing_of_awkward_trees(monkeypatch):
    """One- and two-triangle trees (a leaf for a root), leaves of 40 coincident triangles, flat geometry, a deep
    chain, signed zeros, soups -- and a caller's tree whose boxes do not nest (nothing may be folded there)."""
    rng = np.random.default_rng(5)
    cases = {
        "one": rng.random((1, 9)), "two": rng.random((2, 9)), "three": rng.random((3, 9)),
        "same": np.tile(rng.random((1, 9)), (40, 1)),
        "flat": np.c_[rng.random((500, 2)), np.zeros(500), rng.random((500, 2)), np.zeros(500), rng.random((500, 2)), np.zeros(500)],
        "chain": np.array([[x, -1.0, 4.0, 1.6 * x + 0.01, -1.0, 4.0, 1.3 * x, 1.0, 4.0] for x in 2.2 ** np.arange(56) * 0.02]),
    }
    zeros = rng.random((300, 9)) - 0.5
    zeros[rng.random((300, 9)) < 0.3] = 0.0
    zeros[rng.random((300, 9)) < 0.15] = -0.0
    cases["signed zeros"] = zeros
    for n in (7, 65, 4097, 50_000):
        cases["soup %d" % n] = rng.normal(size=(n, 9)) * np.array([3, 1, 0.2] * 3) + np.repeat(rng.normal(size=(n, 3)) * 4, 3, axis=1).reshape(n, 9)
    for name, verts in cases.items():
        scene = ca.HostScene(vertices=np.ascontiguousarray(verts, dtype=np.float64))
        ctxs = _three_ways(scene.desc, monkeypatch)
        try:
            _same_scene(ctxs, name)
        finally:
            for c in ctxs.values():
                c.close()

    # boxes that do not nest + a plane beyond binary32: the fold check and the bound are per node, on the device too
    scene = ca.HostScene(vertices=np.ascontiguousarray(rng.normal(size=(3000, 9)) * 0.1 + np.tile(rng.normal(size=(3000, 3)) * 5, 3), dtype=np.float64))
    d = scene.desc
    plain = ca.GpuContext(0)
    plain.upload(d)
    folded_as_built = plain.scene_info().n_quads
    plain.close()
    nodes = (ca.Node * d.n_nodes)()
    C.memmove(nodes, d.nodes, C.sizeof(ca.Node) * d.n_nodes)
    inner = [i for i in range(1, d.n_nodes) if nodes[i].count == 0]
    for i in rng.choice(inner, size=len(inner) // 4, replace=False):
        for a in range(3):
            lo, hi = nodes[i].bmin[a], nodes[i].bmax[a]
            nodes[i].bmin[a], nodes[i].bmax[a] = lo + 0.25 * (hi - lo), hi - 0.25 * (hi - lo)
    edited = ca.SceneDesc(d.n_triangles, d.n_nodes, d.n_lights, 0, d.vertices, d.materials, nodes, d.indexes, d.lights)
    ctxs = _three_ways(edited, monkeypatch, build=False)
    try:
        _same_scene(ctxs, "boxes that do not nest")
        assert ctxs["device"].scene_info().n_quads > folded_as_built + 10          # shrunken nodes head QuadNodes of their own
    finally:
        for c in ctxs.values():
            c.close()
    nodes[0].bmax[1] = 1e35
    ctxs = _three_ways(edited, monkeypatch, build=False)
    try:
        _same_scene(ctxs, "a plane at 1e35")
        assert all(np.isinf(b) for b in ctxs["device"].scene_info().plane_bound)
    finally:
        for c in ctxs.values():
            c.close()


def test_nan_vertex_is_refused_by_the_device_build_only():
    """The parallel box fold is proven for NaN-free corners (ct_build.hip): a mesh with a NaN vertex must be refused
    by the device build -- loudly -- and still upload with the host builder's tree."""
    rng = np.random.default_rng(2)
    verts = rng.random((200, 9))
    verts[17, 4] = np.nan
    scene = ca.HostScene(vertices=verts)
    ctx = ca.GpuContext(0)
    try:
        with pytest.raises(RuntimeError, match="NaN"):
            ctx.upload(scene.desc, build_on_device=True)
        with pytest.raises(RuntimeError, match="no scene"):
            ctx.scene_info()
        ctx.upload(scene.desc)
        assert ctx.scene_info().n_triangles == 200
    finally:
        ctx.close()


def test_contexts_share_one_scene(scenes):
    """ctgpu_share_scene: frames in flight on several contexts, ONE copy of the scene in HBM; it outlives its owner."""
    scene = scenes("bunny")
    owner, a, b = ca.GpuContext(0), ca.GpuContext(0), ca.GpuContext(0)
    try:
        with pytest.raises(RuntimeError, match="no scene"):
            a.share_scene(owner)
        owner.upload(scene.desc, build_on_device=True)
        a.share_scene(owner)
        b.share_scene(a)
        assert owner.scene_info().shared_by == 3
        frame = scene.frame(640, 360, rng_seed=8)
        want, _ = owner.render(frame)
        fa, fb = a.render_begin(frame), b.render_begin(frame)          # two frames in flight over the one scene
        assert a.render_poll(wait=True)[0] and b.render_poll(wait=True)[0]
        assert (fa == want).all() and (fb == want).all()
        owner.close()
        assert a.scene_info().shared_by == 2
        assert (a.render(frame)[0] == want).all()
        a.upload(scenes("import").desc)                                # a moves on to another scene; b keeps the bunny
        assert b.scene_info().shared_by == 1 and (b.render(frame)[0] == want).all()
    finally:
        for c in (owner, a, b):
            c.close()


def test_vertices_to_first_frame(scenes):
    """The start-up path the reference runs at RayThread's first call (raythread.cpp:641-654), dragon stand-in:
    vertices in host memory -> tree built and packed in HBM -> first 1080p frame on the host (done in round 2:
    build on the device, arrays to the host, serial fold, 490 MB of copies: 0.36 s)."""
    scene = scenes("dragon")
    frame = scene.frame(1920, 1080, rng_seed=0)
    warm = ca.GpuContext(0)                   # (the process's first HIP calls and kernel loads are not this path's cost)
    warm.upload(scenes("import").desc, build_on_device=True)
    warm.render(scenes("import").frame(64, 64))
    ctx = ca.GpuContext(0)
    try:
        t0 = time.perf_counter()
        ctx.upload(scene.desc, build_on_device=True)
        t1 = time.perf_counter()
        fb, st = ctx.render(frame)
        t2 = time.perf_counter()
        info = ctx.scene_info()
        print("dragon: upload + BuildBVH + packing %.1f ms (device: build %.2f, packing %.2f), first frame %.1f ms, together %.1f ms"
              % ((t1 - t0) * 1e3, info.build_ms, info.pack_ms, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
        assert (fb != 0).any()
        assert (t2 - t0) < 0.25                       # measured ~60 ms; a generous bound for a noisy box
    finally:
        ctx.close()
        warm.close()
