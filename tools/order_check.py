#!/usr/bin/env python3
"""Heavy chunks first (ctgpu_order_info): kernel times of the same frame with the order off and on, frames compared.
    python tools/order_check.py [scene ...]      (W, H, CTGPU_ORDER_HEAVY, CTGPU_ORDER_SPLIT_BELOW from the environment)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))


def context(scene, order):
    old = os.environ.get("CTGPU_ORDER")
    os.environ["CTGPU_ORDER"] = "1" if order else "0"
    try:
        ctx = ca.GpuContext(0)
    finally:
        if old is None:
            del os.environ["CTGPU_ORDER"]
        else:
            os.environ["CTGPU_ORDER"] = old
    ctx.upload(scene.desc, build_on_device=True)
    return ctx


for name in (sys.argv[1:] or ["dragon", "bunny"]):
    scene = ca.HostScene(os.path.join(A, "scene_import.json" if name == "import" else "scene_import_%s.json" % name))
    off, on = context(scene, False), context(scene, True)
    for ss in (False, True):
        f = scene.frame(W, H, supersampling=ss)
        frames = {}
        for label, ctx in (("off", off), ("on", on)):
            for _ in range(3):
                ctx.render(f)
            ctx.kernel_timing(True)
            ctx.read_kernel_times(); ctx.read_walk_work()
            n, ms = 10, 0.0
            for _ in range(n):
                fb, st = ctx.render(f); ms += st.kernel_ms
            t = ctx.read_kernel_times(); w = ctx.read_walk_work()
            ctx.kernel_timing(False)
            frames[label] = (fb, st.rays_primary, st.rays_shadow, st.rays_shadow_elided)
            per = {k: t.total_ms[i] / max(1, t.launches[i]) for i, k in enumerate(t.NAMES)}
            print("%s ss=%d order %-3s kernel %.3f ms  primary %.3f  shadow %.3f  (grid %.3f gate %.3f shade %.3f)  fetches/frame %.0f + %.0f  %s" %
                  (name, ss, label, ms / n, per["ct_walk_packet<primary>"], per["ct_walk_packet<shadow>"], per["ct_sample_grid"], per["ct_light_gate"],
                   per["ct_shade"], (w.quad_nodes[0] + w.triangle_records[0]) / n, (w.quad_nodes[1] + w.triangle_records[1]) / n,
                   ("lists %s split %s" % ctx.order_info()) if label == "on" else ""), flush=True)
        same = (frames["off"][0] == frames["on"][0]).all() and frames["off"][1:] == frames["on"][1:]
        print("%s ss=%d frames and ray counts %s" % (name, ss, "identical" if same else "DIFFER"), flush=True)
        if not same:
            sys.exit(1)
    off.close(); on.close()
