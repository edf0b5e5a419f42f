#!/usr/bin/env python3
"""Host start-up by stage: scene parse + mesh import, then triangle gather + BVH build + packing.
CT_LOAD_THREADS / CT_BVH_THREADS choose the thread counts (1 = the reference's serial way); CT_LOAD_TIMING=1
prints the PLY loader's own stage times."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
lib = ca.host_library()
A = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
for name in sys.argv[1:] or ["dragon"]:
    for rep in range(3):
        t0 = time.time()
        h = lib.cth_load_scene(os.path.join(A, "scene_import_%s.json" % name).encode())
        t1 = time.time()
        d = ca.SceneDesc()
        lib.cth_scene_desc(h, C.byref(d))
        t2 = time.time()
        print("%s: parse+import %.3f s, gather+bvh+pack %.3f s (CT_LOAD_THREADS=%s CT_BVH_THREADS=%s)" %
              (name, t1 - t0, t2 - t1, os.environ.get("CT_LOAD_THREADS", "auto"), os.environ.get("CT_BVH_THREADS", "auto")))
        lib.cth_free_scene(h)
