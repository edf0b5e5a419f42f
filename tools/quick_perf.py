#!/usr/bin/env python3
"""Quick GPU timing of the BASELINE.json configs (not the bench contract; see bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
names = sys.argv[1:] or ["import", "bunny", "dragon"]
W, H = (int(v) for v in os.environ.get("SIZE", "1920x1080").split("x"))      # (SIZE=3840x2160: BASELINE configs[4])
files = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json",
         "dragon": "scene_import_dragon.json"}
for name in names:
    t0 = time.time()
    scene = ca.HostScene(os.path.join(A, files[name]))
    d = scene.desc
    t1 = time.time()
    ctx = ca.GpuContext(0)
    ctx.upload(d)
    t2 = time.time()
    print("%s: %d tris, %d nodes, depth %d; parse+bvh %.2fs, upload %.2fs" %
          (name, d.n_triangles, d.n_nodes, scene.lib.cth_tree_depth(scene.handle), t1 - t0, t2 - t1))
    for ss in (False, True):
        frame = scene.frame(W, H, supersampling=ss)
        best = None
        for _ in range(4):
            fb, st = ctx.render(frame)
            if best is None or st.kernel_ms < best.kernel_ms:
                best = st
        print("  %dx%d ss=%d: rays %d (p %d s %d)  kernel %.3f ms  total %.3f ms  -> %.1f Mrays/s (kernel)" %
              (W, H, ss, best.rays, best.rays_primary, best.rays_shadow, best.kernel_ms, best.total_ms,
               best.rays / best.kernel_ms / 1e3))
    ctx.close()
