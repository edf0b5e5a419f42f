#!/usr/bin/env python3
"""Shadow packets formed by depth (ctgpu_config.shadow_split_gap): shadow-walk time, packets and fetches per gap value,
frames compared with the first.     python tools/shadow_split_sweep.py [scene ...] [--gaps=-1,0.01,0.05]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
files = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json", "dragon": "scene_import_dragon.json"}
gaps = [-1.0, 0.0, 0.005, 0.02, 0.05, 0.2]
for a in sys.argv[1:]:
    if a.startswith("--gaps"): gaps = [float(x) for x in a.split("=")[1].split(",")]
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["dragon", "bunny"]
for name in names:
    scene = ca.HostScene(os.path.join(A, files[name]))
    ref = {}
    for gap in gaps:
        cfg = ca.default_config()
        cfg.shadow_split_gap = gap
        ctx = ca.GpuContext(0, config=cfg)
        ctx.upload(scene.desc, build_on_device=True)
        for ss in (True, False):
            frame = scene.frame(1920, 1080, supersampling=ss)
            for _ in range(3): fb, st = ctx.render(frame)
            ctx.read_walk_work(); ctx.kernel_timing(True); ctx.read_kernel_times()
            n = 8
            for _ in range(n): fb, st = ctx.render(frame)
            kt = ctx.read_kernel_times(); ctx.kernel_timing(False); w = ctx.read_walk_work()
            same = "" if ss not in ref else ("  differing pixels vs first: %d" % int((fb != ref[ss]).sum()))
            if ss not in ref: ref[ss] = fb.copy()
            print("%-7s gap %6.3f ss=%d: shadow walk %.3f ms (primary %.3f)  shadow packets %8d  quads %9d  records %9d  rays %9d%s" %
                  (name, gap, ss, kt.total_ms[3] / max(1, kt.launches[3]), kt.total_ms[1] / max(1, kt.launches[1]), w.packets[1] // n,
                   w.quad_nodes[1] // n, w.triangle_records[1] // n, st.rays_shadow, same), flush=True)
        ctx.close()
