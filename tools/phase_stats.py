#!/usr/bin/env python3
"""Per-phase divergence statistics of ct_walk (needs the CT_WALK_STATS diagnostic build via CTGPU_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["CTGPU_DEBUG_PHASES"] = "1"
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % (sys.argv[1] if len(sys.argv) > 1 else "dragon")))
ctx = ca.GpuContext(0); ctx.upload(scene.desc)
for ss in (False, True):
    f = scene.frame(1920, 1080, supersampling=ss)
    fb, st = ctx.render(f)
    print("ss=%d rays %d kernel %.2f ms" % (ss, st.rays, st.kernel_ms))
