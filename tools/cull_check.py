#!/usr/bin/env python3
"""The leaf cull (triangle boxes, CTGPU_BIG_LEAF) against the walk without it: same pixels, kernel times, fetch counts.
     python tools/cull_check.py [scene ...] [--mins 0,5,9]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
files = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json", "dragon": "scene_import_dragon.json"}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
mins = [0, 8]
SIZE = (1920, 1080)
for a in sys.argv[1:]:
    if a.startswith("--mins"): mins = [int(x) for x in a.split("=")[1].split(",")]
    if a.startswith("--share"): os.environ["CTGPU_BIG_LEAF_SHARE"] = a.split("=")[1]
    if a.startswith("--size"): SIZE = tuple(int(x) for x in a.split("=")[1].split("x"))
for name in (args or ["import", "bunny", "dragon"]):
    scene = ca.HostScene(os.path.join(A, files[name]))
    ref = {}
    for big in mins:
        os.environ["CTGPU_BIG_LEAF"] = str(big)
        ctx = ca.GpuContext(0)
        ctx.upload(scene.desc, build_on_device=True)
        info = ctx.scene_info()
        n_with_cull = len(ctx.scene_array(6)) // 128
        print("%-7s big_leaf>=%-2d: %d triangles, %d reference nodes, %d QuadNodes, %d cull nodes, stack bound %d, %.1f MB resident" %
              (name, big, info.n_triangles, info.n_nodes, info.n_quads, max(0, n_with_cull - info.n_quads), info.quad_pending, info.hbm_bytes / 1e6), flush=True)
        for ss in (True, False):
            frame = scene.frame(SIZE[0], SIZE[1], supersampling=ss)
            best = None
            for _ in range(5):
                fb, st = ctx.render(frame)
                if best is None or st.kernel_ms < best.kernel_ms: best = st
            work = ctx.read_walk_work()
            ctx.kernel_timing(True); ctx.read_kernel_times()
            for _ in range(5): ctx.render(frame)
            kt = ctx.read_kernel_times(); ctx.kernel_timing(False)
            ctx.read_walk_work()
            walks = "primary %.3f shadow %.3f" % (kt.total_ms[1] / max(1, kt.launches[1]), kt.total_ms[3] / max(1, kt.launches[3]))
            same = "" if (ss not in ref) else ("  differing pixels vs first: %d" % int((fb != ref[ss]).sum()))
            if ss not in ref: ref[ss] = fb.copy()
            print("%-7s big_leaf>=%-2d ss=%d: kernel %7.3f ms (%s)  rays %9d  quads p/s %9d %9d  records p/s %9d %9d%s" %
                  (name, big, ss, best.kernel_ms, walks, best.rays, work.quad_nodes[0] // 5, work.quad_nodes[1] // 5,
                   work.triangle_records[0] // 5, work.triangle_records[1] // 5, same), flush=True)
        ctx.close()
