"""-m gpu: cull nodes (cobbletrace_amd/csrc/ct_device.h, DESIGN appendix D) -- the small trees of triangle boxes that stand
in for the reference's many-triangle leaves in the tree supersampled frames walk.  The reference tests every triangle of a
leaf it enters (bvh.cpp:208-218); a lane may leave out the tests whose outcome the grown box proves.  Here: the structure
(every box holds its triangles with the promised margin, the triangles keep the leaf's order), bit-exact t / index /
any-hit against the oracle on rays chosen to sit on the boxes' and triangles' edges, and which scenes get cull nodes at all."""
import os

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu

QUAD = np.dtype([("plane", "<f4", (4, 3, 2)), ("ref", "<u4", (4,)), ("count", "<u4"), ("pad", "<u4", (3,))])
TRI = np.dtype([("p1", "<f8", (3,)), ("e1", "<f8", (3,)), ("e2", "<f8", (3,)), ("triangle", "<u4"), ("remaining", "<u4")])
LEAF, ONE, SLOT = 0x80000000, 0x40000000, 0x3FFFFFFF


def _oracle():
    from oracle import oracle
    return oracle


def _context(monkeypatch, scene, **env):
    """a context for the scene under exactly these knobs (the suite is also run with knobs in the environment: a test
    that is ABOUT a knob owns it -- the shared walk, the default leaf thresholds unless given)"""
    for name in ("CTGPU_PACKET", "CTGPU_BIG_LEAF", "CTGPU_BIG_LEAF_SHARE", "CTGPU_PACK"):
        monkeypatch.delenv(name, raising=False)
    env.setdefault("CTGPU_PACKET", 1)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc)
    for k in env:
        monkeypatch.delenv(k)
    return ctx          # (the caller's environment is back when the test ends: monkeypatch)


def _cull_arrays(ctx):
    info = ctx.scene_info()
    plain = ctx.scene_array(1).view(QUAD)
    cull = ctx.scene_array(6).view(QUAD)
    tris = ctx.scene_array(5).view(TRI)
    return info, plain, cull, tris


def _grown_box(tris, slots):
    """binary64 box of the triangles at `slots`, each grown by 1/256 of its own extent per axis (appendix D: 2 eta ext, eta = 2^-9)"""
    t = tris[slots]
    v = np.stack([t["p1"], t["p1"] + t["e1"], t["p1"] + t["e2"]], axis=1)          # [n, 3 vertices, 3 axes]
    lo, hi = v.min(1), v.max(1)
    pad = (hi - lo) / 256.0
    return (lo - pad).min(0), (hi + pad).max(0)


def _check_subtree(cull, tris, node, n_plain, slots_out, depth=0):
    q = cull[node]
    assert 1 <= q["count"] <= 4 and depth < 8
    for k in range(q["count"]):
        ref = int(q["ref"][k])
        mine = []
        if ref & LEAF:
            assert ref & ONE, "a cull node's leaf entries name ONE triangle"
            mine.append(ref & SLOT)
        else:
            assert (ref & ONE) and (ref & SLOT) >= n_plain, "a cull node only refers to cull nodes, marked as its entries"
            _check_subtree(cull, tris, ref & SLOT, n_plain, mine, depth + 1)
        lo, hi = _grown_box(tris, np.array(mine))
        assert (q["plane"][k, :, 0].astype(np.float64) <= lo).all() and (q["plane"][k, :, 1].astype(np.float64) >= hi).all(), (node, k)
        slots_out.extend(mine)
    for k in range(q["count"], 4):          # absent entries: one-point boxes
        assert (q["plane"][k, :, 0] == q["plane"][k, :, 1]).all()


def test_cull_nodes_hold_their_triangles_in_order(scenes, monkeypatch):
    """scene_import.json (54-triangle leaves): the tree with cull nodes is the plain folded tree with every big leaf's
    reference replaced by a subtree whose boxes hold their triangles, grown, and whose triangles come in the leaf's order."""
    scene = scenes("import")
    ctx = _context(monkeypatch, scene, CTGPU_BIG_LEAF=8)
    try:
        info, plain, cull, tris = _cull_arrays(ctx)
        n = info.n_quads
        assert len(plain) == n and len(cull) > n, "scene_import.json gets cull nodes"
        replaced = 0
        for q in range(n):
            assert plain[q]["count"] == cull[q]["count"] and (plain[q]["plane"] == cull[q]["plane"]).all()
            for k in range(plain[q]["count"]):
                a, b = int(plain[q]["ref"][k]), int(cull[q]["ref"][k])
                if a == b:
                    if a & LEAF:
                        assert tris[a & SLOT]["remaining"] < 8, "a leaf of 8 or more was left alone"
                    continue
                assert (a & LEAF) and not (b & (LEAF | ONE)) and b >= n
                first, count = a & SLOT, int(tris[a & SLOT]["remaining"])
                assert count >= 8
                slots = []
                _check_subtree(cull, tris, b, n, slots)
                assert slots == list(range(first, first + count)), "triangles in the leaf's own order"
                replaced += 1
        assert replaced >= 10
        # every cull node is reachable exactly once
        assert sum(1 for q in cull[n:] if q["count"] > 0) == len(cull) - n
    finally:
        ctx.close()


def _clustered_scene(rng, clusters=6, per_cluster=120, scale=1.0):
    """triangles whose centroids coincide per cluster (BuildBVH cannot split them: bvh.cpp:95-96 gives up when a side is
    empty) -> leaves of `per_cluster` triangles, several cull levels; sizes from tiny to cluster-sized, some degenerate"""
    tris = []
    for c in range(clusters):
        centre = rng.normal(size=3) * 4.0
        for i in range(per_cluster):
            size = 10.0 ** rng.uniform(-2.5, 0.3)
            a, b = rng.normal(size=3) * size, rng.normal(size=3) * size
            if i % 37 == 0:
                b = a * 0.5                                   # a degenerate (zero-area) triangle
            # centroid = (p1 + p2 + p3) * 0.3333f exactly at the centre: p3 = -(p1 + p2) about it
            p1, p2 = centre + a, centre + b
            p3 = centre - a - b
            tris.append(np.concatenate([p1, p2, p3]))
    return np.array(tris) * scale


def _edge_rays(scene, n, seed):
    """rays aimed where the cull's decisions are closest: at vertices, along edges, at box corners of single triangles,
    from inside the geometry, with the usual degenerate ones mixed in"""
    rng = np.random.default_rng(seed)
    v = scene.vertices().reshape(-1, 3, 3)
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    span = (hi - lo).max()
    rays = np.zeros(n, dtype=ca.bindings.RAY_DTYPE)
    tri = v[rng.integers(0, len(v), n)]
    w = rng.random((n, 3))
    kind = rng.integers(0, 6, n)
    w[kind == 0] = np.eye(3)[rng.integers(0, 3, (kind == 0).sum())]                     # exactly a vertex
    w[kind == 1, 2] = 0.0                                                                 # on the edge p1-p2
    w[kind == 2, 0] = 0.0                                                                 # on the edge p2-p3
    w /= np.maximum(w.sum(1, keepdims=True), 1e-300)
    target = (tri * w[:, :, None]).sum(1)
    corner = kind == 3                                                                    # a corner of the triangle's own box
    pick = rng.integers(0, 2, (n, 3)).astype(bool)
    target[corner] = np.where(pick[corner], tri[corner].max(1), tri[corner].min(1))
    origin = (lo + hi) / 2 + rng.normal(size=(n, 3)) * span
    inside = kind == 4
    origin[inside] = (lo + rng.random((n, 3)) * (hi - lo))[inside]
    direction = target - origin
    beyond = kind == 5                                                                    # the triangle lies BEHIND the origin
    direction[beyond] *= -1.0
    direction *= 10.0 ** rng.uniform(-3, 3, (n, 1))                                       # unnormalised, as shadow rays are
    rays["origin"], rays["direction"], rays["t"] = origin, direction, np.float32(1e30)
    rays["direction"][: n // 60, rng.integers(0, 3)] = 0.0
    rays["t"][n // 60: n // 30] = 0.0                                                     # reflected-ray starts (never skip)
    rays["t"][n // 30: n // 24] = np.float32(span)                                        # a caller's own start (never skip)
    return rays


def _same_hits(ctx, scene, rays, what):
    want = _oracle().trace_rays(scene.desc, rays)
    got = ctx.trace_rays_packet(rays)
    for field in ("found", "triangle"):
        assert (got[field] == want[field]).all(), (what, field, int((got[field] != want[field]).sum()))
    assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all(), what
    assert (got["tclosest"].view(np.uint32) == want["tclosest"].view(np.uint32)).all(), what
    fresh = rays.copy()
    fresh["t"] = np.float32(1e30)
    assert (ctx.trace_rays_packet(fresh, any_hit=True)["found"] == _oracle().trace_rays(scene.desc, fresh)["found"]).all(), what
    return want


@pytest.mark.parametrize("scale", [1.0, 1e-3, 300.0])
def test_rays_on_the_edges_of_the_boxes(scale, monkeypatch):
    """Clusters of 120 triangles around one centroid each (one leaf per cluster, four cull levels), at three scales (the
    largest beyond what the gate of appendix D admits: such a scene gets no cull nodes): closest hit and any-hit of 150 k
    rays aimed at vertices, edges and box corners, bit for bit the oracle's."""
    rng = np.random.default_rng(int(scale * 1000) % 9973)
    scene = ca.HostScene(vertices=_clustered_scene(rng, scale=scale))
    ctx = _context(monkeypatch, scene, CTGPU_BIG_LEAF=8)
    try:
        info, plain, cull, tris = _cull_arrays(ctx)
        if scale <= 1.0:
            assert len(cull) - info.n_quads >= 100, "the clusters' leaves got their cull nodes"
        else:
            assert len(cull) == 0, "magnitudes the gate never opens for: no cull nodes"
        want = _same_hits(ctx, scene, _edge_rays(scene, 150_000, seed=3), "clusters x%g" % scale)
        assert want["found"].sum() > 10_000
        # the same packets without cull nodes: the other instantiation of the walk agrees, too
        other = _context(monkeypatch, scene, CTGPU_BIG_LEAF=0)
        try:
            assert len(other.scene_array(6)) == 0
            _same_hits(other, scene, _edge_rays(scene, 20_000, seed=4), "no cull nodes")
        finally:
            other.close()
    finally:
        ctx.close()


def test_rays_through_the_import_scene(scenes, monkeypatch):
    scene = scenes("import")
    ctx = _context(monkeypatch, scene, CTGPU_BIG_LEAF=4)
    try:
        want = _same_hits(ctx, scene, _edge_rays(scene, 200_000, seed=9), "import")
        assert want["found"].sum() > 40_000
    finally:
        ctx.close()


def test_which_scenes_get_cull_nodes(scenes, monkeypatch):
    """Only where they pay (cobbletrace_amd/csrc/ct_pack.hip): a fifth of the triangles in leaves of 8 or more, and
    magnitudes the gate can ever open for.  The bunny (leaves of 6 at most) and the cube scene (200-unit floor triangles)
    keep their plain trees -- and the walk kernels that do not carry the cull code."""
    for name, expected in (("import", True), ("bunny", False), ("cube", False)):
        ctx = _context(monkeypatch, scenes(name))
        try:
            assert (len(ctx.scene_array(6)) > 0) == expected, name
        finally:
            ctx.close()


@pytest.mark.parametrize("ss", [True, False])
def test_frames_with_and_without_cull_nodes(ss, scenes, monkeypatch):
    """scene_import.json 640x360 against the oracle with cull nodes from 4 triangles up and without any; the walk's own
    count of triangle records shows the cull at work in the supersampled frame (a single-sample frame walks the plain tree)."""
    scene = scenes("import")
    frame = scene.frame(640, 360, supersampling=ss, rng_seed=3)
    want, _ = _oracle().render(scene.desc, frame)
    records = {}
    for big in (4, 0):
        # (the count below is of the walks of the reference's tree, where the cull nodes live: the trees of appendix E / F off)
        ctx = _context(monkeypatch, scene, CTGPU_BIG_LEAF=big, CTGPU_SHADOW_TREE=0, CTGPU_CLOSEST_TREE=0)
        try:
            ctx.read_walk_work()
            fb, st = ctx.render(frame)
            assert (fb == want).all(), (big, int((fb != want).sum()))
            assert st.cull_tree == (1 if big and ss else 0) and st.walk_kind == ca.bindings.WALK_PACKET
            records[big] = sum(ctx.read_walk_work().triangle_records)
        finally:
            ctx.close()
    if ss:
        assert records[4] * 4 < records[0], records
    else:
        assert records[4] == records[0], records
