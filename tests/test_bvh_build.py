"""Structure of the BVH the host builder returns (bvh.cpp:16-121 semantics)."""
import numpy as np
import pytest

import cobbletrace_amd as ca

NODE = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("left", "<u4"), ("first", "<u4"), ("count", "<u4"), ("pad", "<u4")])


def tree(scene):
    return scene.nodes().copy().view(NODE).ravel(), scene.indexes().copy(), scene.vertices().copy()


@pytest.mark.parametrize("name", ["cube", "import", "bunny"])
def test_tree_invariants(name, scenes):
    nodes, idx, verts = tree(scenes(name))
    n = len(idx)
    assert sorted(idx.tolist()) == list(range(n))                       # a permutation
    seen = np.zeros(n, dtype=int)
    inner = nodes[nodes["count"] == 0]
    leaves = nodes[nodes["count"] > 0]
    assert len(inner) + len(leaves) == len(nodes) and len(leaves) == len(inner) + 1
    for leaf in leaves:
        seen[leaf["first"]:leaf["first"] + leaf["count"]] += 1
    assert (seen == 1).all()                                            # every triangle in exactly one leaf
    # children are allocated in consecutive pairs after their parent (DFS numbering)
    assert (inner["left"] > 0).all() and (inner["left"] % 2 == 1).all()
    assert len(set(inner["left"].tolist())) == len(inner)
    # leaf bounds are exactly the min / max of their vertices (starting from +-1e30f)
    for leaf in leaves[:: max(1, len(leaves) // 500)]:
        v = verts[idx[leaf["first"]:leaf["first"] + leaf["count"]]].reshape(-1, 3)
        assert (leaf["bmin"] == v.min(0)).all() and (leaf["bmax"] == v.max(0)).all()
    # a parent's box is the union of its children's
    for k in np.flatnonzero(nodes["count"] == 0)[:: max(1, len(inner) // 500)]:
        l, r = nodes[nodes[k]["left"]], nodes[nodes[k]["left"] + 1]
        assert (nodes[k]["bmin"] == np.minimum(l["bmin"], r["bmin"])).all()
        assert (nodes[k]["bmax"] == np.maximum(l["bmax"], r["bmax"])).all()


def test_split_rule_on_a_hand_made_case():
    """Four triangles along x: the root splits at the binary32 midpoint of its extent on the
    longest axis; centroids are scaled by (double)0.3333f, not 1/3."""
    tris = []
    for x in (0.0, 1.0, 2.0, 10.0):
        tris.append([x, 0, 0, x + 0.5, 0, 0, x, 0.5, 0])
    scene = ca.HostScene(vertices=np.array(tris))
    nodes, idx, _ = tree(scene)
    root = nodes[0]
    assert root["count"] == 0 and root["left"] == 1
    left, right = nodes[1], nodes[2]
    # split plane = 0 + 10.5 * 0.5 = 5.25: three triangles left, one right
    assert (left["count"], right["count"]) == (0, 1) or (left["count"], right["count"]) == (3, 1)
    assert idx[right["first"]] == 3
    assert scene.desc.n_nodes == 7 or scene.desc.n_nodes == 5


def test_degenerate_inputs():
    one = ca.HostScene(vertices=np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0.0]]))
    assert one.desc.n_nodes == 1
    same = ca.HostScene(vertices=np.tile(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0.0]]), (9, 1)))
    nodes, _, _ = tree(same)
    assert same.desc.n_nodes == 1 and nodes[0]["count"] == 9           # identical centroids cannot be split


def test_builder_is_deterministic(scenes, assets):
    import os
    a = scenes("import")
    b = ca.HostScene(os.path.join(assets, "scene_import.json"))
    assert a.lib.cth_node_hash(a.handle) == b.lib.cth_node_hash(b.handle)
    assert a.lib.cth_index_hash(a.handle) == b.lib.cth_index_hash(b.handle)
    assert a.lib.cth_triangle_hash(a.handle) == b.lib.cth_triangle_hash(b.handle)


@pytest.mark.parametrize("name", ["bunny", "dragon"])
def test_threaded_build_returns_the_same_tree(name, monkeypatch):
    """Meshes of 50,000 triangles and more are built on several threads (top of the tree on one,
    subtrees in parallel, slots numbered afterwards in the reference's allocation order): node for
    node and index for index the result is the plain recursion's."""
    import os
    import __graft_entry__ as entry
    entry.prepare_assets(need_dragon=(name == "dragon"))
    path = os.path.join(os.path.dirname(__file__), "..", "assets", "scene_import_%s.json" % name)
    built = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("CT_BVH_THREADS", threads)
        scene = ca.HostScene(path)
        built[threads] = (scene.nodes().copy(), scene.indexes().copy())
        scene.close()
    assert built["1"][0].shape[0] > 80000
    for threads in ("3", "8"):
        assert built[threads][0].tobytes() == built["1"][0].tobytes()
        assert (built[threads][1] == built["1"][1]).all()
