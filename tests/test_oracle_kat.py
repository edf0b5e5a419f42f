"""The oracle (and the product's loader / BVH builder feeding it) against the known answers of
the compiled reference recorded in SURVEY.md (tests/golden/reference_kat.json): triangle and node
counts, leaf histogram, parsed constants, individual pixel values, hit-pixel counts, ray counts
and the gprof call counts of IntersectAABB / IntersectTriangle.  These are the facts that pin the
oracle to the reference; the survey's framebuffer hashes are not reproducible (see the JSON)."""
import collections
import json
import os

import numpy as np
import pytest

import cobbletrace_amd as ca
from oracle import oracle

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))


def test_scene_counts(scenes):
    for name, want in KAT["scenes"].items():
        scene = scenes(name)
        d = scene.desc
        assert d.n_triangles == want["triangles"]
        assert d.n_nodes == want["nodes"]
        if "objects" in want:
            assert scene.lib.cth_object_count(scene.handle) == want["objects"]


def test_leaf_structure_matches_reference(scenes):
    bunny = scenes("bunny")
    counts = bunny.nodes()[:, 56:60].copy().view(np.uint32).ravel()
    hist = collections.Counter(counts[counts > 0].tolist())
    assert {str(k): v for k, v in hist.items()} == KAT["scenes"]["bunny"]["leaf_histogram"]
    assert bunny.lib.cth_tree_depth(bunny.handle) == KAT["scenes"]["bunny"]["depth_frames"]
    imp = scenes("import")
    counts = imp.nodes()[:, 56:60].copy().view(np.uint32).ravel()
    assert counts.max() == KAT["scenes"]["import"]["max_leaf"]
    p1 = bunny.vertices()[0, :3]
    assert p1.tolist() == KAT["scenes"]["bunny"]["first_triangle_p1"]          # bit-exact doubles


def test_parsed_constants(scenes):
    cam = scenes("import").camera
    assert list(cam) == KAT["parsed_constants"]["import_camera"]
    lights = scenes("bunny").desc.lights
    assert abs(lights[0].intensity - KAT["parsed_constants"]["float_0.2"]) < 1e-9
    assert lights[0].intensity == np.float32(0.2) and lights[1].intensity == np.float32(0.6)


@pytest.mark.parametrize("case", KAT["frames"], ids=lambda c: "%s-T%d" % (c["scene"], c["threads"]))
def test_reference_frames(case, scenes):
    scene = scenes(case["scene"])
    frame = scene.frame(case["width"], case["height"], threads=case["threads"], supersampling=case["supersampling"])
    count = "aabb_tests" in case
    fb, st = oracle.render(scene.desc, frame, os_threads=8, count_work=count)
    for x, y, value in case.get("pixels", []):
        assert fb[y, x] == int(value, 16), "pixel (%d,%d)" % (x, y)
    if "non_background_pixels" in case:
        assert int(((fb != 0x333333) & (fb != 0)).sum()) == case["non_background_pixels"]
    if "primary_rays" in case:
        assert st.rays_primary == case["primary_rays"]
        assert st.rays_shadow == case["shadow_rays"] == 2 * case["hits"]
        assert st.rays == case["traversals"]
        assert st.node_visits == case["aabb_tests"]            # 67,022,780 IntersectAABB calls
        assert st.triangle_tests == case["triangle_tests"]     # 5,535,094 IntersectTriangle calls
    if "zero_rows" in case:
        lo, hi = case["zero_rows"]
        assert (fb[0] == 0).all() and (fb[lo:hi + 1] == 0).all()
        assert (fb[case["first_rendered_row"]] != 0).any()
        assert (fb[:, :420] == 0).all() and (fb[:, 1500:] == 0).all()


def test_bvhtest_known_answer():
    """bvh.cpp:223-310: 256 rand() triangles, product builder, any-hit walk of 640x640 rays."""
    import ctypes as C
    lib = oracle.library()
    verts = np.zeros(256 * 9)
    lib.cto_bvhtest_triangles(verts.ctypes.data)
    scene = ca.HostScene(vertices=verts)
    assert scene.desc.n_nodes == KAT["bvhtest"]["nodes"]
    fb = np.zeros((640, 640), dtype=np.uint32)
    hits = C.c_uint64()
    assert lib.cto_bvhtest_render(C.byref(scene.desc), fb.ctypes.data, C.byref(hits)) == 0
    assert hits.value == KAT["bvhtest"]["hit_pixels"]
    assert set(np.unique(fb).tolist()) == {0, int(KAT["bvhtest"]["pixel_value"], 16)}
