#!/usr/bin/env python3
"""Queue chunk sizes (ctgpu_config.chunk / chunk_shadow) on the headline frame: per-walk times.
     python tools/chunk_sweep.py [scene]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
files = {"import": "scene_import.json", "bunny": "scene_import_bunny.json", "dragon": "scene_import_dragon.json"}
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene = ca.HostScene(os.path.join(A, files[name]))
frame = scene.frame(1920, 1080)
for rep in range(2):
    for chunk, chunk_shadow in ((0, 0), (128, 256), (128, 512), (128, 1024), (256, 256), (256, 512), (64, 128)):
        cfg = ca.default_config()
        cfg.chunk, cfg.chunk_shadow = chunk, chunk_shadow
        ctx = ca.GpuContext(0, config=cfg)
        ctx.upload(scene.desc, build_on_device=True)
        for _ in range(4): ctx.render(frame)
        ctx.kernel_timing(True); ctx.read_kernel_times()
        n = 12
        for _ in range(n): ctx.render(frame)
        kt = ctx.read_kernel_times(); ctx.kernel_timing(False)
        ms = [kt.total_ms[k] / max(1, kt.launches[k]) for k in range(5)]
        print("%s chunk %4d shadow %4d: frame %.3f primary %.3f shadow %.3f" % (name, chunk, chunk_shadow, sum(ms), ms[1], ms[3]), flush=True)
        ctx.close()
