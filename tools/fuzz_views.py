#!/usr/bin/env python3
"""Randomised views of the shipped meshes (deep trees, sub-pixel triangles): GPU frames and single rays vs the oracle.
    python tools/fuzz_views.py [seconds] [first_seed] [scene ...]
Cameras around, close to and inside the mesh's bounding box, random orientations, both sampling modes, random
thread counts / sub-sampling; 16k rays per view through the general kernel and the frame kernel (closest and any hit)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
names = sys.argv[3:] or ["bunny", "dragon", "import"]
entry.prepare_assets(need_dragon="dragon" in names)
A = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
scenes = {}
for name in names:
    s = ca.HostScene(os.path.join(A, "scene_import.json" if name == "import" else "scene_import_%s.json" % name))
    c = ca.GpuContext(0)
    c.upload(s.desc)
    v = s.vertices().reshape(-1, 3)
    scenes[name] = (s, c, v.min(0), v.max(0))
t_end = time.time() + budget
views = rays_total = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    name = names[seed % len(names)]
    scene, ctx, lo, hi = scenes[name]
    centre, span = (lo + hi) / 2, float((hi - lo).max())
    distance = float(rng.choice([0.0, 0.05, 0.6, 2.0, 6.0]))          # 0: inside the mesh's box
    cam = centre + rng.normal(size=3) * span * 0.25 + np.array([0, 0, -1.0]) * span * distance
    angles = [np.float32(a) for a in rng.uniform(-0.5, 0.5, 3)]
    w, h = int(rng.choice([96, 160, 320])), int(rng.choice([64, 120, 240]))
    frame = scene.frame(w, h, threads=int(rng.choice([1, 4, 8, 32])) if h >= 32 else 1, supersampling=bool(rng.random() < 0.5),
                        subsampling=bool(rng.random() < 0.15), rotation=oracle.camera_rotation(*angles), rng_seed=seed)
    frame.cam_pos = (C.c_double * 3)(*cam)
    # (the context carries the chunk costs of its last frame -- another view, often another size -- into this one, and
    #  this frame's into the second render: heavy chunks first, on stale and on fresh history)
    got, _ = ctx.render(frame)
    again, _ = ctx.render(frame)
    want, _ = oracle.render(scene.desc, frame, os_threads=16)
    n_diff = int((got != want).sum()) + int((again != want).sum())
    m = 16384
    rays = np.zeros(m, dtype=ca.bindings.RAY_DTYPE)
    rays["origin"] = cam + rng.normal(size=(m, 3)) * span * rng.choice([0.0, 0.001, 0.3])
    target = lo + rng.random((m, 3)) * (hi - lo)
    rays["direction"] = target - rays["origin"]
    rays["direction"][: m // 32, int(rng.integers(3))] = 0.0
    rays["t"] = np.float32(1e30)
    ref = oracle.trace_rays(scene.desc, rays)
    same = True
    for hits in (ctx.trace_rays(rays), ctx.trace_rays_packet(rays)):
        same &= bool((hits["found"] == ref["found"]).all() and (hits["triangle"] == ref["triangle"]).all() and
                     (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
    same &= bool((ctx.trace_rays_packet(rays, any_hit=True)["found"] == ref["found"]).all())
    views += 1
    rays_total += 3 * m
    print("seed %d %s %dx%d T=%d ss=%d sub=%d distance %.2f: %d px differ, %d hit px, rays %s" %
          (seed, name, w, h, frame.num_threads, frame.flags & 1, (frame.flags >> 1) & 1, distance, n_diff,
           int(((want != 0x333333) & (want != 0)).sum()), "equal" if same else "DIFFER"), flush=True)
    if n_diff or not same:
        print("MISMATCH at seed %d" % seed)
        sys.exit(1)
    seed += 1
print("%d views, %d rays: every frame bit-identical to the oracle's, every ray equal" % (views, rays_total))
