#!/usr/bin/env python3
"""Section statistics of ct_walk_packet (needs the CT_WALK_STATS diagnostic build via CTGPU_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["CTGPU_DEBUG_PACKET"] = "1"
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
for name in (sys.argv[1:] or ["dragon"]):
    scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
    ctx = ca.GpuContext(0); ctx.upload(scene.desc)
    f = scene.frame(int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), supersampling=os.environ.get("SS", "1") != "0")
    fb, st = ctx.render(f)
    print("%s rays %d kernel %.2f ms" % (name, st.rays, st.kernel_ms))
    ctx.close()
