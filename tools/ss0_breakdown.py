#!/usr/bin/env python3
"""Where a single-sample frame's time goes: per-kernel event times and walk work, ss off (and on for comparison)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
for name in (sys.argv[1:] or ["dragon", "bunny"]):
    scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
    ctx = ca.GpuContext(0); ctx.upload(scene.desc)
    for ss in (False, True):
        f = scene.frame(W, H, supersampling=ss)
        for _ in range(3): ctx.render(f)
        ctx.kernel_timing(True)
        ctx.read_kernel_times(); w0 = ctx.read_walk_work()
        n = 10
        ms = 0.0
        for _ in range(n):
            fb, st = ctx.render(f); ms += st.kernel_ms
        t = ctx.read_kernel_times(); w = ctx.read_walk_work()
        ctx.kernel_timing(False)
        print("%s ss=%d rays %d (primary %d shadow %d elided %d) kernel %.3f ms" % (name, ss, st.rays, st.rays_primary,
              st.rays_shadow, st.rays_shadow_elided, ms / n))
        for i, k in enumerate(t.NAMES):
            print("   %-26s %8.3f ms" % (k, t.total_ms[i] / max(1, t.launches[i])))
        for j, k in enumerate(("primary", "shadow")):
            p = (w.packets[j]) / n; q = w.quad_nodes[j] / n; r = w.triangle_records[j] / n
            print("   %-8s packets %9.0f  quads/packet %6.1f  records/packet %6.1f" % (k, p, q / max(p, 1), r / max(p, 1)))
    ctx.close()
