"""-m gpu: the shadow tree (cobbletrace_amd/csrc/ct_device.h "the shadow tree", ct_shadow_tree.hip, DESIGN appendix E) -- the
tree of this library's own that the frame's shadow rays walk instead of the reference's.  A shadow ray only asks whether
any triangle the reference tests moves ray.t (raythread.cpp:303-307), so the order of the tests is free; what is NOT free is
which triangles count: only those in leaves the reference enters.  Here: the structure as it sits in HBM, frames and ray
counts against the walk of the reference's own tree and against the oracle, the rays the tree does not take (zero
direction components), and a caller's tree whose leaf boxes were shrunk so that the triangle test and the reference's
leaf-box test disagree on many rays -- the walk must then follow the box, as the reference does."""
import ctypes as C

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu

QUAD = np.dtype([("plane", "<f4", (4, 3, 2)), ("ref", "<u4", (4,)), ("count", "<u4"), ("axis", "<u4"), ("pad", "<u4", (2,))])
TRI = np.dtype([("p1", "<f8", (3,)), ("e1", "<f8", (3,)), ("e2", "<f8", (3,)), ("triangle", "<u4"), ("remaining", "<u4")])
NODE = np.dtype([("bmin", "<f8", (3,)), ("bmax", "<f8", (3,)), ("left", "<u4"), ("first", "<u4"), ("count", "<u4"), ("pad", "<u4")])
LEAF, ONE, SLOT = 0x80000000, 0x40000000, 0x3FFFFFFF


def _oracle():
    from oracle import oracle
    return oracle


def _context(shadow_tree, desc, build=0, leaf=2, fill=50, closest=0, order=None):
    """a context that owns the knobs this file is about (the suite may run with CTGPU_SHADOW_TREE* in the environment)"""
    cfg = ca.config_from_environment()
    cfg.shadow_tree, cfg.shadow_tree_build, cfg.shadow_tree_leaf, cfg.shadow_tree_leaf_fill = shadow_tree, build, leaf, fill
    cfg.closest_tree = closest
    if order is not None:
        cfg.order = order
    cfg.packet_walk = 1
    ctx = ca.GpuContext(0, cfg)
    ctx.upload(desc)
    return ctx


@pytest.mark.parametrize("name", ["import", "bunny"])
@pytest.mark.parametrize("build,leaf,fill", [(0, 2, 50), (0, 1, 0), (1, 2, 50), (2, 4, 60)])
def test_the_tree_in_hbm(name, build, leaf, fill, scenes):
    """Every triangle under exactly one entry, every entry's box around what lies below it, a leaf's entry around its
    triangles grown by appendix D's 1/256 and by gamma; the tree's own records are the reference's, permuted, leaf by leaf;
    entries a node does not have are the corner point with a reference that is safe to follow; a node's entries stand in
    ascending order of their centres along the axis the node names (the closest-hit walk goes through them forwards or
    backwards, as its rays point), so any entry may have the node's count - 1 others waiting: the stack bound the library
    reports is the tree's under that rule; leaf_of_slot names the reference's leaves."""
    scene = scenes(name)
    ctx = _context(2, scene.desc, build, leaf, fill)
    try:
        info = ctx.scene_info()
        assert info.shadow_tree_nodes > 0 and info.shadow_tree_gamma > 0
        quads = ctx.scene_array(7).view(QUAD).copy()
        # (in HBM a reference is the BYTE OFFSET of what it names -- a node: index * 128, a leaf: LEAF | first record * 80 --
        #  so that a visit's scalar loads take it as it is; back to indices for the checks)
        refs = quads["ref"]
        leafs = (refs & LEAF) != 0
        # (... and a leaf's reference carries, in the four low bits a multiple of 16 leaves free, its number of triangles - 1:
        #  the walk notes a leaf's tests without fetching a record)
        sizes = np.where(leafs, (refs & 15) + 1, 0)
        refs = np.where(leafs, refs & ~np.uint32(15), refs)
        assert ((refs[leafs] & ~np.uint32(LEAF)) % 80 == 0).all() and (refs[~leafs] % 128 == 0).all()
        quads["ref"] = np.where(leafs, LEAF | ((refs & ~np.uint32(LEAF)) // 80), refs // 128)
        tris = ctx.scene_array(9).view(TRI)
        reference_order = ctx.scene_array(5).view(TRI)
        leaf_of = ctx.scene_array(8).view(np.uint32)
        n = info.n_triangles
        assert len(quads) == info.shadow_tree_nodes and len(leaf_of) == n and len(tris) == n
        # the same records, permuted (the triangle's index names it), each leaf a run that counts down
        # (the tree's own records carry, where the reference's carry the scene triangle, their slot in the REFERENCE's order:
        #  what hits[] names and what decides between equal t -- ct_walk_closest_tree)
        assert sorted(tris["triangle"].tolist()) == list(range(n))
        same = reference_order[tris["triangle"]]
        for f in ("p1", "e1", "e2"):
            assert (same[f] == tris[f]).all()
        scene_triangle = same["triangle"]
        rem = tris["remaining"]
        assert rem.min() >= 1 and rem.max() <= leaf and (rem[:-1][rem[:-1] > 1] - 1 == rem[1:][rem[:-1] > 1]).all() and rem[-1] == 1
        starts = np.concatenate([[True], rem[:-1] == 1])
        v = np.stack([tris["p1"], tris["p1"] + tris["e1"], tris["p1"] + tris["e2"]], axis=1)
        lo, hi = v.min(1), v.max(1)
        pad = (hi - lo) / 256.0 + info.shadow_tree_gamma * 0.999
        seen_tri = np.zeros(n, bool)
        seen_node = np.zeros(len(quads), bool)
        box_lo = np.zeros((len(quads), 3)); box_hi = np.zeros((len(quads), 3))
        stack = [(0, 0)]
        worst = 0
        order = []
        while stack:
            at, waiting = stack.pop()
            assert not seen_node[at]
            seen_node[at] = True
            order.append(at)
            q = quads[at]
            count = int(q["count"])
            assert 1 <= count <= 4
            worst = max(worst, waiting + count - 1)
            axis = int(q["axis"])
            assert axis < 3 and (q["pad"] == 0).all()
            centres = q["plane"][:count, axis, :].astype(np.float64).sum(1) * 0.5
            assert (np.diff(centres) >= -4.0 * info.shadow_tree_gamma - 1e-6 * np.abs(centres).max()).all(), (at, centres)
            for k in range(count):
                ref = int(q["ref"][k])
                if ref & LEAF:
                    first = ref & SLOT
                    assert first < n and starts[first] and int(sizes[at][k]) == int(rem[first])
                    for t in range(first, first + int(rem[first])):
                        assert not seen_tri[t]
                        seen_tri[t] = True
                        assert (q["plane"][k][:, 0] <= lo[t] - pad[t]).all() and (q["plane"][k][:, 1] >= hi[t] + pad[t]).all()
                else:
                    assert at < ref < len(quads)
                    stack.append((ref, waiting + count - 1))
            for k in range(count, 4):
                assert (q["plane"][k][:, 0] == q["plane"][k][:, 1]).all()
                assert (int(q["ref"][k]) & LEAF) and (int(q["ref"][k]) & SLOT) < n and starts[int(q["ref"][k]) & SLOT]
        assert seen_tri.all() and seen_node.all()
        assert worst == info.shadow_tree_pending <= 63
        for at in reversed(order):          # children come after their parents in the array
            q = quads[at]
            count = int(q["count"])
            box_lo[at] = q["plane"][:count, :, 0].min(0); box_hi[at] = q["plane"][:count, :, 1].max(0)
            for k in range(count):
                ref = int(q["ref"][k])
                if not ref & LEAF:
                    assert (q["plane"][k][:, 0] <= box_lo[ref]).all() and (q["plane"][k][:, 1] >= box_hi[ref]).all()
        # the corner the absent entries sit at lies beyond every plane
        present = np.concatenate([quads["plane"][i, :int(quads["count"][i])].ravel() for i in range(0, len(quads), max(len(quads) // 2000, 1))])
        absent = [quads["plane"][i, int(quads["count"][i]):].ravel() for i in range(len(quads)) if quads["count"][i] < 4]
        if absent:
            assert np.concatenate(absent).min() >= np.abs(present).max()
        # leaf_of_slot against the reference's own node array
        nodes = scene.nodes().view(NODE).reshape(-1)
        indexes = scene.indexes()
        leaf_of_triangle = np.full(n, 0xFFFFFFFF, np.uint32)
        for i in np.nonzero(nodes["count"])[0]:
            leaf_of_triangle[indexes[nodes["first"][i]:nodes["first"][i] + nodes["count"][i]]] = i
        assert (leaf_of == leaf_of_triangle[scene_triangle]).all()
    finally:
        ctx.close()


def test_which_scenes_get_a_shadow_tree(scenes):
    """Appendix D's lemma needs K K S <= 2^20: the cube scene's 200-unit floor triangles are beyond it, and that scene keeps
    walking the reference's tree (same pixels either way); a caller's tree whose boxes do not nest gets none either (the
    leaf's box would no longer speak for its ancestors')."""
    for name, expected in (("cube", False), ("import", True), ("bunny", True)):
        ctx = _context(1, scenes(name).desc)
        try:
            assert (ctx.scene_info().shadow_tree_nodes > 0) == expected, name
        finally:
            ctx.close()
    scene = scenes("import")
    d = scene.desc
    nodes = (ca.Node * d.n_nodes)()
    C.memmove(nodes, d.nodes, C.sizeof(ca.Node) * d.n_nodes)
    inner = next(i for i in range(1, d.n_nodes) if nodes[i].count == 0)
    nodes[nodes[inner].left].bmax[0] += 10.0                 # a child sticking out of its parent
    copy = ca.SceneDesc(d.n_triangles, d.n_nodes, d.n_lights, 0, d.vertices, d.materials, nodes, d.indexes, d.lights)
    ctx = _context(1, copy)
    try:
        assert ctx.scene_info().shadow_tree_nodes == 0
        frame = scene.frame(240, 136, threads=8, supersampling=True, rng_seed=2)
        got, _ = ctx.render(frame)
        want, _ = _oracle().render(copy, frame, os_threads=2)
        assert (got == want).all()
    finally:
        ctx.close()
    off = _context(0, scene.desc)
    try:
        assert off.scene_info().shadow_tree_nodes == 0        # CTGPU_SHADOW_TREE=0: not even built
    finally:
        off.close()


@pytest.mark.parametrize("name,size", [("import", (640, 360)), ("bunny", (640, 360)), ("dragon", (480, 270))])
def test_frames_and_counts_equal_the_reference_trees_walk(name, size, scenes):
    """The same frame with the shadow rays through the shadow tree (both builds) and through the reference's tree: pixels,
    ray counts, the number of rays that found a blocker.  Supersampled and single-sample (CTGPU_SHADOW_TREE=2)."""
    scene = scenes(name)
    ref = _context(0, scene.desc)
    trees = [_context(2, scene.desc, 0, 2, 50), _context(2, scene.desc, 1, 1, 0), _context(2, scene.desc, 2, 4, 70)]
    try:
        for ss in (True, False):
            frame = scene.frame(size[0], size[1], supersampling=ss, rng_seed=5)
            ref.read_walk_work()
            want, wst = ref.render(frame)
            wwork = ref.read_walk_work()
            for ctx in trees:
                ctx.read_walk_work()
                got, st = ctx.render(frame)
                work = ctx.read_walk_work()
                assert st.shadow_tree == 1 and wst.shadow_tree == 0 and st.rays_shadow_left == 0        # (ctgpu_stats says which code ran)
                assert (got == want).all(), "%s ss=%d: %d pixels differ" % (name, ss, int((got != want).sum()))
                assert (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (wst.rays_primary, wst.rays_shadow, wst.rays_shadow_elided)
                assert work.shadow_rays_blocked == wwork.shadow_rays_blocked
                assert work.packets[1] > 0 and work.triangle_records[1] > 0
        # and against the oracle (the supersampled frame, a band of it)
        frame = scene.frame(size[0], size[1], supersampling=True, rng_seed=5)
        got, _ = trees[0].render(frame)
        bands = frame.num_threads
        want, _ = _oracle().render(scene.desc, frame, os_threads=4, band_first=bands // 2, band_count=1)
        rows = np.nonzero((want != 0).any(axis=1))[0]
        assert len(rows) and (got[rows] == want[rows]).all()
    finally:
        ref.close()
        for ctx in trees:
            ctx.close()


@pytest.mark.parametrize("name,size", [("bunny", (640, 360)), ("dragon", (480, 270)), ("import", (640, 360))])
def test_primary_rays_through_the_tree(name, size, scenes):
    """CTGPU_CLOSEST_TREE: the primary rays walk the shadow tree's nodes too (appendix F) -- the least t over everything the
    reference could test, among equal ones the triangle it meets first, the winner checked against the reference's own box of
    its leaf, everything else left to the reference-tree walk.  Frames, ray counts and what ctgpu_stats says, against the walk
    of the reference's tree; a band against the oracle.  (Frames that hand their heavy chunks out first keep the
    reference's tree for their primary rays: the contexts here switch that order off so that small frames take the walk.)
    The import scene's hits lie on axis-aligned flat faces, whose leaf boxes are flat: every such winner sits on the rim of
    its box, where tmin and t are two roundings of one number -- the winner's check asks for tmin below what the reference
    can CARRY there (the least t of any other tested triangle, appendix F.1), not below t itself, and takes them."""
    scene = scenes(name)
    ref = _context(0, scene.desc, order=0)
    tree = _context(2, scene.desc, closest=2, order=0)
    try:
        for ss in (True, False):
            frame = scene.frame(size[0], size[1], supersampling=ss, rng_seed=11)
            want, wst = ref.render(frame)
            assert wst.closest_tree == 0 and wst.rays_primary_left == 0
            for k in range(2):
                got, st = tree.render(frame)
                assert (got == want).all(), "%s ss=%d frame %d: %d pixels differ" % (name, ss, k, int((got != want).sum()))
                assert (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (wst.rays_primary, wst.rays_shadow, wst.rays_shadow_elided)
                assert st.closest_tree == 1 and st.shadow_tree == 1 and st.rays_primary_left * 100 <= st.rays_primary, (st.rays_primary_left, st.rays_primary)
        frame = scene.frame(size[0], size[1], supersampling=True, rng_seed=11)
        fresh = _context(2, scene.desc, closest=2, order=0)
        try:
            got, st = fresh.render(frame)
            assert st.closest_tree == 1
        finally:
            fresh.close()
        bands = frame.num_threads
        want, _ = _oracle().render(scene.desc, frame, os_threads=4, band_first=bands // 2, band_count=1)
        rows = np.nonzero((want != 0).any(axis=1))[0]
        assert len(rows) and (got[rows] == want[rows]).all()
    finally:
        ref.close(); tree.close()


def test_a_context_that_went_back_looks_again(tmp_path):
    """A scene that hands most of its primary rays back -- a caller's tree whose leaf boxes are half their size: three winners of
    four lie outside -- pays both walks; its context keeps the reference's tree after the first frame that left more than a
    sixteenth (`closest_off`) and tries the tree again every 256 frames, in case the camera now looks at something else.
    The oracle's pixels (for the tree as given) on every frame of the way."""
    scene, copy = _sheet_with_shrunk_leaf_boxes(tmp_path, seed=8)
    tree = _context(2, copy, closest=2, order=0)
    try:
        frame = scene.frame(192, 108, threads=4, supersampling=True, rng_seed=1)
        want, _ = _oracle().render(copy, frame, os_threads=4)
        took = []
        for k in range(1, 260):
            got, st = tree.render(frame)
            assert (got == want).all(), k
            if st.closest_tree:
                took.append(k)
                assert st.rays_primary_left * 16 > st.rays_primary
        assert took == [1, 257], took
    finally:
        tree.close()


def test_an_eye_far_away_keeps_the_references_tree(scenes):
    """The boxes' growth covers the binary32 arithmetic for origins within o_max = 2 V of the scene (appendix E.1); a shadow
    ray starts on a surface, a primary ray at the eye -- wherever the caller puts it.  A camera beyond that (checked by the
    host, per frame) keeps the reference's tree for its primary rays; its shadow rays still start on surfaces and walk the
    shadow tree.  Moved back inside, the next frame's primary rays take the tree again.  Rotated cameras as well."""
    scene = scenes("bunny")
    tree = _context(2, scene.desc, closest=2, order=0)
    try:
        near = scene.frame(480, 270, supersampling=True, rng_seed=6)
        far = scene.frame(480, 270, supersampling=True, rng_seed=6)
        extent = float(np.abs(np.ctypeslib.as_array(scene.desc.vertices, shape=(scene.desc.n_triangles * 9,))).max())
        far.cam_pos = (C.c_double * 3)(near.cam_pos[0] * 0.5, near.cam_pos[1], -(40.0 * extent + 50.0))
        turned = scene.frame(480, 270, supersampling=True, rng_seed=6, rotation=ca.camera_rotation(0.21, -0.13, 0.05))
        for frame, expected in ((near, 1), (far, 0), (turned, 1), (near, 1)):
            got, st = tree.render(frame)
            want, _ = _oracle().render(scene.desc, frame, os_threads=4)
            assert (got == want).all(), int((got != want).sum())
            assert st.closest_tree == expected and st.shadow_tree == 1
    finally:
        tree.close()


def test_equal_t_goes_to_the_triangle_the_reference_meets_first(tmp_path):
    """bvh.cpp:212 records a triangle only when it lowers ray.t STRICTLY: of several triangles hit at the same t the one the
    reference's walk meets first wins, and with it its colour.  Stacks of coincident triangles in different colours, in a
    scene with enough other geometry that the tree's order and the reference's differ: the tree's walk keeps the least
    {t, place in the reference's order}."""
    rng = np.random.default_rng(5)
    tris = _bumpy_sheet(rng, n=10)
    colours = [[250, 40, 40], [40, 250, 40], [40, 40, 250], [240, 240, 40], [40, 240, 240]]
    coloured = []
    for k in range(40):
        c = np.array([rng.uniform(-1.6, 1.6), rng.uniform(-0.5, 0.8), rng.uniform(1.5, 4.0)])
        a, b = rng.standard_normal(3) * 0.5, rng.standard_normal(3) * 0.5
        p = [c.tolist(), (c + a).tolist(), (c + b).tolist()]
        for j in rng.permutation(len(colours))[:int(rng.integers(2, 5))]:
            coloured.append((p, colours[j]))
    # ... and stacks that lie in axis-aligned planes, one triangle a leaf or two: the leaf's box is flat, tmin and t are two
    # roundings of one number, and with a second triangle at the very same t the winner's check cannot take the hit (the
    # least t of any OTHER tested triangle is t itself): those rays are left, and still come out right
    for k in range(24):
        c = np.array([rng.uniform(-1.6, 1.6), rng.uniform(-0.5, 0.8), rng.uniform(1.5, 4.0)])
        axis = k % 3
        a, b = np.zeros(3), np.zeros(3)
        a[(axis + 1) % 3], b[(axis + 2) % 3] = rng.uniform(0.3, 0.7), rng.uniform(0.3, 0.7)
        p = [c.tolist(), (c + a).tolist(), (c + b).tolist()]
        for j in rng.permutation(len(colours))[:int(rng.integers(1, 4))]:
            coloured.append((p, colours[j]))
    import json
    objs = [{"type": "triangle", "p1": t[0], "p2": t[1], "p3": t[2], "color": [200, 160, 90], "specular": 40, "reflection": 0} for t in tris]
    objs += [{"type": "triangle", "p1": p[0], "p2": p[1], "p3": p[2], "color": col, "specular": -1, "reflection": 0} for p, col in coloured]
    order = rng.permutation(len(objs))
    scene_json = {"objects": [objs[i] for i in order], "camera": {"position": [0, 0.8, -2.5]},
                  "lights": [{"type": "ambient", "intensity": 0.6}, {"type": "directional", "intensity": 0.4, "direction": [-1, 2, -1.5]}],
                  "settings": {"numberOfThreads": 4, "subsampling": False, "wireframe": False, "supersampling": True}}
    path = tmp_path / "stacks.json"
    path.write_text(json.dumps(scene_json))
    scene = ca.HostScene(str(path))
    tree = _context(2, scene.desc, closest=2, order=0)
    try:
        for ss in (True, False):
            frame = scene.frame(320, 240, threads=4, supersampling=ss, rng_seed=2)
            got, st = tree.render(frame)
            want, _ = _oracle().render(scene.desc, frame, os_threads=4)
            assert st.closest_tree == 1 and st.rays_primary_left > 0
            assert (got == want).all(), "%d pixels differ" % int((got != want).sum())
            assert len(np.unique(want)) > 6                    # several of the stacks' colours are in the picture
    finally:
        tree.close()


def test_heavy_chunks_first_through_the_shadow_tree(scenes):
    """A single-sample 1080p frame has few packets per wave and hands its heavy chunks out first, in parts (ctgpu_order_info);
    with CTGPU_SHADOW_TREE=2 its shadow rays walk the shadow tree, whose walk reads the same lists and writes the same cost
    words: same pixels and counts as the unordered walk of the reference's tree, and lists that fill up."""
    scene = scenes("dragon")
    cfg = ca.config_from_environment()
    cfg.shadow_tree, cfg.order, cfg.packet_walk = 0, 0, 1
    plain = ca.GpuContext(0, cfg); plain.upload(scene.desc)
    cfg = ca.config_from_environment()
    cfg.shadow_tree, cfg.order, cfg.packet_walk = 2, 1, 1
    tree = ca.GpuContext(0, cfg); tree.upload(scene.desc)
    try:
        frame = scene.frame(1920, 1080, supersampling=False)
        want, wst = plain.render(frame)
        for k in range(4):
            got, st = tree.render(frame)
            assert st.ordered == 1 and st.shadow_tree == 1 and st.walk_kind == ca.bindings.WALK_PACKET
            assert (got == want).all(), "frame %d: %d pixels differ" % (k, int((got != want).sum()))
            assert (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (wst.rays_primary, wst.rays_shadow, wst.rays_shadow_elided)
        entries, split = tree.order_info()
        assert split and sum(entries[:4]) > 0 and sum(entries[4:]) > 0, entries
    finally:
        plain.close(); tree.close()


def _write_scene(tmp_path, name, triangles, lights, camera):
    import json
    objs = [{"type": "triangle", "p1": t[0], "p2": t[1], "p3": t[2], "color": [200, 160, 90], "specular": 40, "reflection": 0} for t in triangles]
    scene = {"objects": objs, "camera": {"position": list(camera)}, "lights": lights,
             "settings": {"numberOfThreads": 4, "subsampling": False, "wireframe": False, "supersampling": True}}
    path = tmp_path / name
    path.write_text(json.dumps(scene))
    return str(path)


def _bumpy_sheet(rng, n=24):
    """a height field over [-2, 2]^2 with a few blockers hovering above it: shadows of every kind"""
    xs = np.linspace(-2, 2, n + 1)
    h = 0.15 * rng.standard_normal((n + 1, n + 1))
    tris = []
    for i in range(n):
        for j in range(n):
            p = lambda a, b: [float(xs[a]), float(h[a, b]) - 1.0, float(xs[b]) + 3.0]
            tris.append((p(i, j), p(i + 1, j), p(i + 1, j + 1)))
            tris.append((p(i, j), p(i + 1, j + 1), p(i, j + 1)))
    for k in range(12):
        c = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-0.3, 0.6), rng.uniform(1.8, 4.2)])
        a, b = rng.standard_normal(3) * 0.35, rng.standard_normal(3) * 0.35
        tris.append(((c).tolist(), (c + a).tolist(), (c + b).tolist()))
    return tris


def test_rays_the_tree_does_not_take(tmp_path):
    """A directional light along an axis (two zero direction components), one with a single zero component, a point light
    far outside the gate's range for d, and ordinary lights, all in one scene: the rays of the first three are left to the
    second launch over the reference's tree, the others walk the shadow tree; the frame is the oracle's."""
    rng = np.random.default_rng(3)
    lights = [{"type": "ambient", "intensity": 0.1},
              {"type": "directional", "intensity": 0.2, "direction": [0, 1, 0]},
              {"type": "directional", "intensity": 0.2, "direction": [1, 2, 0]},
              {"type": "point", "intensity": 0.2, "position": [10000000, 20000000, -10000000]},
              {"type": "point", "intensity": 0.3, "position": [1.5, 2.5, 1.0]},
              {"type": "directional", "intensity": 0.2, "direction": [-1, 3, -2]}]
    scene = ca.HostScene(_write_scene(tmp_path, "sheet.json", _bumpy_sheet(rng), lights, (0, 0.8, -2.5)))
    tree, ref = _context(2, scene.desc), _context(0, scene.desc)
    try:
        assert tree.scene_info().shadow_tree_nodes > 0
        for ss in (True, False):
            frame = scene.frame(320, 240, threads=4, supersampling=ss, rng_seed=9)
            got, st = tree.render(frame)
            other, ost = ref.render(frame)
            want, cst = _oracle().render(scene.desc, frame, os_threads=4)
            assert (got == want).all() and (other == want).all()
            assert (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (ost.rays_primary, ost.rays_shadow, ost.rays_shadow_elided)
            # the shadow tree took some rays and left others: three of the five lights' rays, to be exact
            work = tree.read_walk_work()
            assert 0 < work.packets[1] and st.shadow_tree == 1 and 0 < st.rays_shadow_left < st.rays_shadow and ost.rays_shadow_left == 0
    finally:
        tree.close(); ref.close()


def _sheet_with_shrunk_leaf_boxes(tmp_path, seed):
    """the bumpy sheet with its own tree, and the same arrays with every LEAF box shrunk to half its size (still nested)"""
    rng = np.random.default_rng(seed)
    lights = [{"type": "ambient", "intensity": 0.15}, {"type": "point", "intensity": 0.5, "position": [1.0, 3.0, 0.5]},
              {"type": "directional", "intensity": 0.3, "direction": [-1, 2, -1.5]}]
    scene = ca.HostScene(_write_scene(tmp_path, "sheet_%d.json" % seed, _bumpy_sheet(rng), lights, (0, 0.8, -2.5)))
    d = scene.desc
    nodes = (ca.Node * d.n_nodes)()
    C.memmove(nodes, d.nodes, C.sizeof(ca.Node) * d.n_nodes)
    for i in range(d.n_nodes):
        if nodes[i].count:
            for a in range(3):
                lo, hi = nodes[i].bmin[a], nodes[i].bmax[a]
                nodes[i].bmin[a] = lo + 0.25 * (hi - lo); nodes[i].bmax[a] = hi - 0.25 * (hi - lo)
    copy = ca.SceneDesc(d.n_triangles, d.n_nodes, d.n_lights, 0, d.vertices, d.materials, nodes, d.indexes, d.lights)
    copy._keep = nodes                  # (the ctypes array must outlive the descriptor)
    return scene, copy


def test_a_triangle_hit_outside_its_leafs_box_does_not_count(tmp_path):
    """The reference tests a triangle only after its leaf's box has passed (bvh.cpp:200-218).  A caller's tree whose LEAF
    boxes are shrunk to half their size (still nested: every box inside its parent's) makes that visible: rays that hit a
    triangle outside the shrunk box of its leaf are not blocked by it in the reference.  The shadow tree knows nothing of
    those boxes; the verdict of the leaf's box after a hit must bring the frame back to the reference's for the tree AS GIVEN."""
    scene, copy = _sheet_with_shrunk_leaf_boxes(tmp_path, seed=8)
    d = scene.desc
    tree, ref = _context(2, copy), _context(0, copy)
    both = _context(2, copy, closest=2, order=0)
    try:
        assert tree.scene_info().shadow_tree_nodes > 0
        for ss in (True, False):
            frame = scene.frame(320, 240, threads=4, supersampling=ss, rng_seed=4)
            got, st = tree.render(frame)
            want, cst = _oracle().render(copy, frame, os_threads=4)
            if ss:      # the primary rays through the tree as well: their winners, too, count only inside the (shrunk) leaf boxes
                again, bst = both.render(frame)
                assert bst.closest_tree == 1 and bst.rays_primary_left > 0
                assert (again == want).all(), "%d pixels differ with the primary rays through the tree" % int((again != want).sum())
            plain, _ = _oracle().render(d, frame, os_threads=4)
            assert (plain != want).any()                      # the shrunk boxes changed what the reference renders
            assert (got == want).all(), "%d pixels differ" % int((got != want).sum())
            other, ost = ref.render(frame)
            assert (other == want).all()
            assert st.rays_shadow_left > 0                        # rays whose hit lay outside the shrunk box of its leaf went on to the second launch
            assert (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (ost.rays_primary, ost.rays_shadow, ost.rays_shadow_elided)
    finally:
        tree.close(); ref.close(); both.close()
