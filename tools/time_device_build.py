#!/usr/bin/env python3
"""Device BVH build (ctgpu_build_bvh) against the host builder: time and identity."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
for name in sys.argv[1:] or ["bunny", "dragon"]:
    scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
    v = scene.vertices()
    best = None
    for rep in range(4):
        t0 = time.time()
        nodes, idx, ms = ca.build_bvh_on_device(v)
        wall = (time.time() - t0) * 1e3
        best = ms if best is None else min(best, ms)
    same = (idx == scene.indexes()).all() and (nodes[:, :60] == scene.nodes().reshape(-1, 64)[:, :60]).all()
    print("%s: %d triangles -> %d nodes; device build %.2f ms (with allocation and copies %.1f ms); identical to the host tree: %s" %
          (name, len(idx), len(nodes), best, wall, same))
