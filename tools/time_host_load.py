#!/usr/bin/env python3
"""Start-up cost on the host: scene parse + mesh import + BVH build (CT_BVH_THREADS=1: plain recursion)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
for name in sys.argv[1:] or ["bunny", "dragon"]:
    for rep in range(3):
        t0 = time.time()
        scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
        print("%s: %d triangles, %d nodes: %.3f s (CT_BVH_THREADS=%s)" %
              (name, scene.desc.n_triangles, scene.desc.n_nodes, time.time() - t0, os.environ.get("CT_BVH_THREADS", "auto")))
        scene.close()
