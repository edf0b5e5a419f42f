#!/usr/bin/env python3
"""Throughput of small frames (what one rank of an 8-GPU job sees) with 1..3 frames in flight."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import cobbletrace_amd as ca
from cobbletrace_amd.multi_gpu import ShardedRenderer
A = os.path.join(os.path.dirname(__file__), "..", "assets")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
full = scene.frame(1920, 1080, supersampling=True)
worlds = [int(w) for w in os.environ.get("OVERLAP_WORLDS", "1,2,4,8").split(",")]
flights = [int(w) for w in os.environ.get("OVERLAP_INFLIGHT", "1,2,3").split(",")]
for world in worlds:
    for inflight in flights:
        # rank 0's share of the real frame: partition (8-row tiles, rank 0 of `world`)
        rs = [ShardedRenderer(scene.desc, 0, world, 0) for _ in range(inflight)]
        ss = [torch.cuda.Stream() for _ in range(inflight)]
        def frame(i):
            with torch.cuda.stream(ss[i % inflight]):
                rs[i % inflight].trace(full)
        for i in range(6): frame(i)
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for i in range(n): frame(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%s world %d inflight %d: %.3f ms per frame-share (x%d = %.3f ms)" % (name, world, inflight, dt * 1e3, world, dt * 1e3 * world))
        for r in rs: r.close()
