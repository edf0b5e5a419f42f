#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity: triangle soups of many sizes and coordinate scales, cameras anywhere (inside the
geometry too), random lights and materials, both sampling modes, frames and single rays through every kernel.
    python tools/fuzz_parity.py [seconds] [first_seed]
Prints one line per scene; exits non-zero at the first difference beyond 1 LSB (frames) or any difference (rays)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import cobbletrace_amd as ca
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = tempfile.mkdtemp()
t_end = time.time() + budget
n_scenes = n_frames = n_rays = 0
worst = 0
n_tree = [0, 0]
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, 7, 40, 300, 2500]))
    scale = float(rng.choice([1e-3, 0.05, 1.0, 30.0, 2000.0]))
    size = float(rng.choice([0.02, 0.3, 2.0]))
    centres = rng.normal(size=(n, 3)) * scale
    # a third of the scenes: triangles in a few clusters around a common centroid each -- BuildBVH cannot split those
    # (bvh.cpp:95-96), so they end up in leaves of dozens or hundreds: the leaves that get cull nodes (DESIGN appendix D)
    clustered = rng.random() < 0.35
    if clustered:
        centres = centres[rng.integers(0, min(n, int(rng.choice([1, 2, 5]))), n)]
    flat = rng.random() < 0.2                                  # axis-aligned flat geometry now and then
    tris = []
    for c in centres:
        p = c[None, :] + rng.normal(size=(3, 3)) * scale * size * (10.0 ** rng.uniform(-2, 0) if clustered else 1.0)
        if clustered:
            p[2] = 3.0 * c - p[0] - p[1]                       # (p1 + p2 + p3) / 3 = c
        if flat:
            p[:, int(rng.integers(3))] = c[int(rng.integers(3))]
        tris.append({"type": "triangle", "p1": [float("%.6g" % v) for v in p[0]], "p2": [float("%.6g" % v) for v in p[1]],
                     "p3": [float("%.6g" % v) for v in p[2]], "color": [int(v) for v in rng.integers(0, 256, 3)],
                     "specular": int(rng.choice([-1, 1, 10, 100, 1000])), "reflection": float(rng.choice([0, 0, 0, 0.3]))})
    # a sixth of the scenes: some triangles once or twice more, vertex for vertex, in another colour and somewhere else in the
    # list -- equal t, and the first the reference's walk meets must win (bvh.cpp:212; DESIGN appendix F.1)
    if rng.random() < 0.17 and tris:
        for i in rng.integers(0, len(tris), max(1, len(tris) // 4)):
            twin = dict(tris[int(i)], color=[int(v) for v in rng.integers(0, 256, 3)])
            tris.insert(int(rng.integers(0, len(tris) + 1)), twin)
    use_spheres = rng.random() < 0.3
    if use_spheres:
        for _ in range(int(rng.integers(1, 6))):
            tris.append({"type": "sphere", "center": [float("%.6g" % v) for v in rng.normal(size=3) * scale],
                         "radius": float("%.4g" % (scale * rng.choice([0.05, 0.4, 1.5]))), "color": [int(v) for v in rng.integers(0, 256, 3)],
                         "specular": int(rng.choice([-1, 10, 500])), "reflection": float(rng.choice([0, 0.4]))})
    lights = [{"type": "ambient", "intensity": float("%.3g" % rng.uniform(0, 0.5))}]
    for _ in range(int(rng.choice([0, 1, 2, 3, 3, 36]))):         # (36: beyond the 32 lights the fast path once stopped at)
        if rng.random() < 0.5:
            lights.append({"type": "point", "intensity": float("%.3g" % rng.uniform(0, 0.8)),
                           "position": [float("%.6g" % v) for v in rng.normal(size=3) * scale * 2]})
        else:
            d = rng.normal(size=3)
            if rng.random() < 0.3:
                d[int(rng.integers(3))] = 0.0                  # an axis-parallel light: zero components in every shadow ray
            lights.append({"type": "directional", "intensity": float("%.3g" % rng.uniform(0, 0.8)), "direction": [float("%.6g" % v) for v in d]})
    cam = rng.normal(size=3) * scale * rng.choice([0.0, 0.5, 3.0])
    cam[2] -= scale * rng.choice([0.0, 2.0, 6.0])
    path = os.path.join(tmp, "fuzz.json")
    json.dump({"objects": tris, "lights": lights, "camera": {"position": [float("%.6g" % v) for v in cam]},
               "settings": {"numberOfThreads": int(rng.choice([1, 3, 8])), "subsampling": bool(rng.random() < 0.2), "wireframe": False,
                            "supersampling": bool(rng.random() < 0.5)}}, open(path, "w"))
    scene = ca.HostScene(path)
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=bool(seed % 2))     # every other scene: tree built and laid out on the GPU
    angles = [np.float32(a) for a in rng.uniform(-0.6, 0.6, 3)] if rng.random() < 0.5 else [np.float32(0)] * 3
    w, h = int(rng.choice([64, 97, 160])), int(rng.choice([48, 64, 121]))
    if len(lights) > 30:
        w, h = 64, 48             # (36 shadow rays per hit, ten bounces, 16 samples: minutes of oracle time on a larger frame)
    frame = scene.frame(w, h, rotation=oracle.camera_rotation(*angles), rng_seed=seed, spheres=use_spheres,
                        reflect_fix=bool(rng.random() < 0.3), rng_mode=(ca.RNG_LIBC if rng.random() < 0.2 else ca.RNG_HASH))
    if frame.rng_mode == ca.RNG_LIBC:
        frame.rng_seed = int(rng.integers(1, 1000))
    got, st = ctx.render(frame)
    # the same frame again, twice: from the second frame on the context hands out heavy chunks first (run this tool with
    # CTGPU_ORDER_HEAVY=0.8,0.6,0.4,0.2 CTGPU_ORDER_FLOOR=1 to have lists on frames this small) -- nothing may change
    for _ in range(2):
        again, st2 = ctx.render(frame)
        # (reflective frames alternate between the bounce levels and the general path while the context measures them:
        #  only the former leaves out the shadow rays whose answer cannot matter, so traced + left out is what agrees)
        if not ((again == got).all() and (st2.rays_primary, st2.rays_shadow + st2.rays_shadow_elided, st2.rays_reflect) ==
                (st.rays_primary, st.rays_shadow + st.rays_shadow_elided, st.rays_reflect)):
            print("MISMATCH between two renders of one frame at seed %d: %d px differ; rays %s then %s; %d triangles %dx%d flags %d rng %d lights %d spheres %s; order %s" %
                  (seed, int((again != got).sum()), (st.rays_primary, st.rays_shadow, st.rays_reflect), (st2.rays_primary, st2.rays_shadow, st2.rays_reflect),
                   n, w, h, frame.flags, frame.rng_mode, len(lights), use_spheres, ctx.order_info()))
            sys.exit(1)
    want, _ = oracle.render(scene.desc, frame, os_threads=4)
    shifts = np.array([0, 8, 16])
    diff = int(np.abs(((got[..., None] >> shifts) & 255).astype(int) - ((want[..., None] >> shifts) & 255).astype(int)).max())
    worst = max(worst, diff)
    ok = diff <= 1 and ((got == 0) == (want == 0)).all()
    # single rays through both kernels
    m = 4096
    rays = np.zeros(m, dtype=ca.bindings.RAY_DTYPE)
    rays["origin"] = cam + rng.normal(size=(m, 3)) * scale * rng.choice([0.0, 0.01, 1.0])
    rays["direction"] = rng.normal(size=(m, 3))
    rays["direction"][: m // 16, int(rng.integers(3))] = 0.0
    rays["t"] = np.float32(1e30)
    rays["t"][m // 2: m // 2 + 64] = 0.0
    ref = oracle.trace_rays(scene.desc, rays)
    ray_ok = True
    for hits in (ctx.trace_rays(rays), ctx.trace_rays_packet(rays)):
        ray_ok &= bool((hits["found"] == ref["found"]).all() and (hits["triangle"] == ref["triangle"]).all() and
                       (hits["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
    fresh = rays.copy()
    fresh["t"] = np.float32(1e30)
    ray_ok &= bool((ctx.trace_rays_packet(fresh, any_hit=True)["found"] == oracle.trace_rays(scene.desc, fresh)["found"]).all())
    ctx_cull = ctx.scene_array(6)
    ctx.close()
    scene.close()
    n_scenes += 1
    n_frames += 3
    n_rays += 3 * m
    n_tree[0] += st.shadow_tree; n_tree[1] += st.closest_tree
    print("seed %d: %d triangles%s%s, scale %g, %dx%d ss=%d sub=%d flags %d rng %d lights %d: max diff %d, %d px differ, rays %s" %
          (seed, n, " (clustered, %d cull nodes)" % (len(ctx_cull) // 128) if clustered else "",
           " [%s]" % "+".join(k for k, on in (("shadow tree", st.shadow_tree), ("primary rays through it", st.closest_tree)) if on) if st.shadow_tree else "", scale, w, h, bool(frame.flags & 1), bool(frame.flags & 2), frame.flags >> 2, frame.rng_mode, len(lights), diff, int((got != want).sum()),
           "equal" if ray_ok else "DIFFER"), flush=True)
    if not ok or not ray_ok:
        print("MISMATCH at seed %d (scene kept in %s)" % (seed, path))
        sys.exit(1)
    seed += 1
print("%d scenes (%d with the shadow rays through the shadow tree, %d with the primary rays too), %d frames, %d rays: no difference beyond %d LSB" %
      (n_scenes, n_tree[0], n_tree[1], n_frames, n_rays, worst))
