#!/usr/bin/env python3
"""A scene several times the benchmark's size through every stage, against the host builder and the oracle:
     python tools/scale_check.py [million triangles = 8]
   triangle soup on a bumpy sheet (every pixel covers many); device BVH build == host build byte for byte; a single-sample
   1920x1080 frame and a supersampled 640x360 one == the oracle's; 200 k rays through the frame kernel == the oracle's."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import cobbletrace_amd as ca
from oracle import oracle

millions = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
side = int(np.sqrt(millions * 1e6 / 2))
n = side * side * 2
rng = np.random.default_rng(3)
t0 = time.time()
# a sheet of side x side cells over x, z in [-8, 8] x [4, 20], height a few waves plus noise; two triangles per cell, jittered
u = np.linspace(-8.0, 8.0, side + 1); w = np.linspace(4.0, 20.0, side + 1)
X, Z = np.meshgrid(u, w, indexing="ij")
Y = -1.5 + 0.4 * np.sin(X * 1.7) * np.cos(Z * 0.9) + rng.normal(size=X.shape) * (8.0 / side)
P = np.stack([X, Y, Z], axis=-1)
a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
verts = np.concatenate([np.concatenate([a, b, c], axis=-1).reshape(-1, 9), np.concatenate([b, d, c], axis=-1).reshape(-1, 9)])
verts = verts[rng.permutation(len(verts))]
print("%d triangles generated in %.1f s" % (len(verts), time.time() - t0)); sys.stdout.flush()

t0 = time.time()
scene = ca.HostScene(vertices=verts)
print("host scene (host BVH build) in %.1f s: %d nodes, depth %d" % (time.time() - t0, len(scene.nodes()), scene.lib.cth_tree_depth(scene.handle))); sys.stdout.flush()
nodes, indexes, ms = ca.build_bvh_on_device(verts)
host_nodes = scene.nodes().reshape(-1, 64)[:, :60]             # bvh_node_t without its 4 bytes of padding
assert len(nodes) == scene.desc.n_nodes and (nodes[:, :60] == host_nodes).all(), "device tree differs from the host builder's"
assert (indexes == scene.indexes()).all(), "device index permutation differs"
print("device BVH build: %.1f ms on the GPU, byte-identical to the host builder's tree" % ms); sys.stdout.flush()

ctx = ca.GpuContext(0)
t0 = time.time(); ctx.upload(scene.desc); print("upload (tree from the host, packed on the device) %.2f s" % (time.time() - t0)); sys.stdout.flush()
fresh = ca.GpuContext(0)
t0 = time.time(); fresh.upload(scene.desc, build_on_device=True); dt = time.time() - t0
info = fresh.scene_info()
print("vertices -> BuildBVH + packing in HBM: %.3f s (device: build %.1f ms, packing %.1f ms); %.0f MB" % (dt, info.build_ms, info.pack_ms, info.hbm_bytes / 1e6))
for which in range(6):
    assert (fresh.scene_array(which) == ctx.scene_array(which)).all(), which
fresh.close()
sys.stdout.flush()
orc = oracle
for (wd, ht, ss) in ((1920, 1080, False), (640, 360, True)):
    f = scene.frame(wd, ht, threads=8, supersampling=ss, rng_seed=1)
    f.cam_pos = (C.c_double * 3)(0.0, 2.5, -4.0)
    for _ in range(6):                      # (the context measures both walks on a scene like this one and keeps the faster)
        got, st = ctx.render(f)
    lane, packet_ms, lane_ms = ctx.walk_choice(ss)
    print("   walk chosen: %s (shared %.2f ms, per-lane %.2f ms)" % ("per-lane" if lane else "shared", packet_ms, lane_ms))
    t0 = time.time(); want, cst = orc.render(scene.desc, f, os_threads=16); dt = time.time() - t0
    diff = int((got != want).sum())
    print("%dx%d ss=%d: %d rays, GPU kernel %.2f ms (%.0f Mrays/s), oracle %.1f s; differing pixels %d, hit pixels %d" % (
          wd, ht, ss, st.rays, st.kernel_ms, st.rays / st.kernel_ms / 1e3, dt, diff, int((want != 0x333333).sum())))
    sys.stdout.flush()
    assert diff == 0
rays = np.zeros(200_000, dtype=ca.bindings.RAY_DTYPE)
rays["origin"] = np.array([0.0, 2.5, -4.0]) + rng.normal(size=(len(rays), 3)) * 0.5
rays["direction"] = np.c_[rng.uniform(-8, 8, len(rays)), rng.uniform(-2.5, -0.5, len(rays)) - 2.5, rng.uniform(4, 20, len(rays)) + 4.0]
rays["t"] = np.float32(1e30)
got = ctx.trace_rays_packet(rays); want = orc.trace_rays(scene.desc, rays)
assert (got["found"] == want["found"]).all() and (got["triangle"] == want["triangle"]).all()
assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
print("200000 rays through the frame kernel: t and triangle bit-equal (%d hits)" % int(want["found"].sum()))
