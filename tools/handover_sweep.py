#!/usr/bin/env python3
"""The hand-over of incoherent packets (ct_walk_packet -> per-lane ct_walk): frame kernel time against the two knobs
   CTGPU_PACKET_BUDGET (visits before a packet is judged) and CTGPU_PACKET_LANES (rays per visit it must average),
   every frame compared with the one rendered with the hand-over switched off.
     python tools/handover_sweep.py [scene ...]     (default: dragon bunny; W, H from the environment)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
GRID = [(0, 12)] + [(b, l) for b in (32, 64, 96, 160) for l in (6, 12, 20, 32)]
if os.environ.get("GRID"):
    GRID = [(0, 12)] + [tuple(int(x) for x in item.split(":")) for item in os.environ["GRID"].split(",")]


def measure(scene, frame, budget, lanes, n=10):
    os.environ["CTGPU_PACKET_BUDGET"], os.environ["CTGPU_PACKET_LANES"] = str(budget), str(lanes)
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=True)
    for _ in range(3):
        fb, st = ctx.render(frame)
    ctx.kernel_timing(True)
    ctx.read_kernel_times()
    ms = 0.0
    for _ in range(n):
        fb, st = ctx.render(frame)
        ms += st.kernel_ms
    t = ctx.read_kernel_times()
    ctx.kernel_timing(False)
    w = ctx.read_walk_work()
    ctx.close()
    per = [t.total_ms[i] / max(1, t.launches[i]) for i in range(5)]
    return fb, st, ms / n, per, w


for name in (sys.argv[1:] or ["dragon", "bunny"]):
    scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
    for ss in (False, True):
        frame = scene.frame(W, H, supersampling=ss)
        base = None
        for budget, lanes in GRID:
            fb, st, ms, per, w = measure(scene, frame, budget, lanes)
            if base is None:
                base = fb
            same = bool((fb == base).all())
            print("%-7s ss=%d budget %3d lanes %2d: kernel %.3f ms  (primary %.3f  shadow %.3f)  %.2f Grays/s  handed over: primary %d shadow %d  %s"
                  % (name, ss, budget, lanes, ms, per[1], per[3], st.rays / ms / 1e6, w.rays_handed_over[0], w.rays_handed_over[1],
                     "same frame" if same else "FRAME DIFFERS"))
            sys.stdout.flush()
            assert same
