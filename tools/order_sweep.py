#!/usr/bin/env python3
"""Heavy chunks first: thresholds x parts over single-sample 1080p frames (kernel ms per frame, fetches per frame).
    python tools/order_sweep.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
SS = bool(int(os.environ.get("SS", "0")))
CASES = [tuple(c.split("/")) for c in os.environ["CASES"].split()] if os.environ.get("CASES") else None
CONFIGS = [None] + [(h, p, pr) for (h, p) in CASES or (("12,12,12,12", "1,1,1,1"), ("4,3,2,1", "4,4,1,1"), ("3,2,1.5,1", "4,1,1,1"), ("5,3,2,1.25", "4,2,1,1"),
           ("3,2,1.5,1", "1,1,1,1"), ("6,3,2,1", "4,1,1,1"), ("8,4,2,1", "4,2,1,1"), ("4,3,2,1", "2,2,1,1"), ("4,2,1.5,1", "8,1,1,1"), ("4,3,2,1.5", "4,4,1,1"), ("5,4,3,2", "4,4,1,1")) for pr in os.environ.get("PRIOS", "0 1").split()]


def context(scene, cfg):
    keys = ("CTGPU_ORDER", "CTGPU_ORDER_HEAVY", "CTGPU_ORDER_PARTS", "CTGPU_ORDER_PRIORITY")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["CTGPU_ORDER"] = "0" if cfg is None else os.environ.get("ORDER_MODE", "1")
    if cfg is not None:
        os.environ["CTGPU_ORDER_HEAVY"] = cfg[0]
        os.environ["CTGPU_ORDER_PARTS"] = cfg[1]
        os.environ["CTGPU_ORDER_PRIORITY"] = cfg[2]
    try:
        ctx = ca.GpuContext(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.upload(scene.desc, build_on_device=True)
    return ctx


for name in (sys.argv[1:] or ["dragon", "bunny"]):
    scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
    f = scene.frame(W, H, supersampling=SS)
    want = None
    for cfg in CONFIGS:
        ctx = context(scene, cfg)
        for _ in range(3):
            ctx.render(f)
        ctx.kernel_timing(True)
        ctx.read_kernel_times(); ctx.read_walk_work()
        n, ms = 10, 0.0
        for _ in range(n):
            fb, st = ctx.render(f); ms += st.kernel_ms
        t = ctx.read_kernel_times(); w = ctx.read_walk_work()
        per = {k: t.total_ms[i] / max(1, t.launches[i]) for i, k in enumerate(t.NAMES)}
        if want is None:
            want = fb
        print("%s ss=%d %-40s kernel %.3f ms  primary %.3f  shadow %.3f  build %.3f  fetches %8.0f + %8.0f  lists %s%s" %
              (name, SS, "off" if cfg is None else "heavy %s parts %s prio %s" % cfg, ms / n, per["ct_walk_packet<primary>"], per["ct_walk_packet<shadow>"],
               per["ct_sample_grid"], (w.quad_nodes[0] + w.triangle_records[0]) / n, (w.quad_nodes[1] + w.triangle_records[1]) / n,
               ctx.order_info()[0], "" if (fb == want).all() else "  FRAME DIFFERS"), flush=True)
        ctx.close()
