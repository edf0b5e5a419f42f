#!/usr/bin/env python3
"""Single-sample frames (1 spp, 1920x1080): per-walk times and fetch counts, frame compared with a reference frame file if given.
     python tools/ss0_check.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
files = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json", "dragon": "scene_import_dragon.json"}
for name in (sys.argv[1:] or ["dragon", "bunny", "import"]):
    scene = ca.HostScene(os.path.join(A, files[name]))
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=True)
    frame = scene.frame(1920, 1080, supersampling=False)
    for _ in range(4): fb, st = ctx.render(frame)
    ctx.read_walk_work(); ctx.kernel_timing(True); ctx.read_kernel_times()
    n, best = 10, 1e9
    for _ in range(n):
        fb, st = ctx.render(frame); best = min(best, st.kernel_ms)
    kt = ctx.read_kernel_times(); ctx.kernel_timing(False); w = ctx.read_walk_work()
    path = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "ss0_%s.npy" % name)
    same = ""
    if os.path.exists(path): same = "  differing pixels vs saved frame: %d" % int((np.load(path) != fb).sum())
    else: np.save(path, fb)
    print("%-7s 1 spp: kernel %.3f ms (best %.3f)  primary %.3f  shadow %.3f   records p/s %d %d  quads p/s %d %d  rays %d%s" %
          (name, sum(kt.total_ms[k] / max(1, kt.launches[k]) for k in range(5)), best, kt.total_ms[1] / max(1, kt.launches[1]),
           kt.total_ms[3] / max(1, kt.launches[3]), w.triangle_records[0] // n, w.triangle_records[1] // n, w.quad_nodes[0] // n,
           w.quad_nodes[1] // n, st.rays, same), flush=True)
    ctx.close()
