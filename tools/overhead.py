#!/usr/bin/env python3
"""Per-frame fixed cost of the pipeline: tiny frames, many iterations (host launch overhead + kernel floors)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import cobbletrace_amd as ca
from cobbletrace_amd.multi_gpu import ShardedRenderer
A = os.path.join(os.path.dirname(__file__), "..", "assets")
scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % (sys.argv[1] if len(sys.argv) > 1 else "bunny")))
r = ShardedRenderer(scene.desc, 0, 1, 0)
for w, h, ss in ((64, 64, False), (1920, 136, True), (1920, 272, True), (1920, 544, True), (1920, 1080, True)):
    f = scene.frame(w, h, threads=1 if h < 100 else 8, supersampling=ss)
    for _ in range(5):
        r.frame(f)
    torch.cuda.synchronize()
    n = 200 if h < 200 else 30
    t0 = time.perf_counter()
    for _ in range(n):
        r.frame(f)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print("%4dx%-4d ss=%d: %.1f us/frame wall, %.1f us/frame host issue" % (w, h, ss, t1 / n * 1e6, t_issue / n * 1e6))
