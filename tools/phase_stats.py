#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["CTGPU_DEBUG_PHASES"] = "1"
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
scene = ca.HostScene(os.path.join(A, "scene_import_dragon.json"))
ctx = ca.GpuContext(0); ctx.upload(scene.desc)
for ss in (False, True):
    f = scene.frame(1920, 1080, supersampling=ss)
    st = ctx.count_work(f)
    print("ss=%d rays %d visits %d tests %d kernel %.2f ms" % (ss, st.rays, st.node_visits, st.triangle_tests, st.kernel_ms))
