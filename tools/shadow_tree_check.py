#!/usr/bin/env python3
"""The shadow tree against the reference's tree, one process, one box: same pixels?  what does the shadow walk cost?
     python tools/shadow_tree_check.py [scene ...] [--size WxH] [--builds 0:2:20,1:1:0 (strategy:leaf:fill)] [--single]
Per scene: a context whose shadow rays walk the reference's tree (CTGPU_SHADOW_TREE=0) and one per build strategy whose
shadow rays walk the shadow tree; frames compared word for word, kernel times by the library's own HIP events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
FILES = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json",
         "dragon": "scene_import_dragon.json", "one": "scene_file_1.json"}
args = sys.argv[1:]
size = (1920, 1080)
builds = [(0, 2, 50)]
single = False
closest = [1]
names = []
i = 0
while i < len(args):
    if args[i] == "--size": size = tuple(int(v) for v in args[i + 1].split("x")); i += 2
    elif args[i] == "--builds": builds = [tuple(int(x) for x in v.split(":")) for v in args[i + 1].split(",")]; i += 2
    elif args[i] == "--single": single = True; i += 1
    elif args[i] == "--closest": closest = [int(v) for v in args[i + 1].split(",")]; i += 2
    else: names.append(args[i]); i += 1
names = names or ["dragon"]
bad = 0


def measure(ctx, frame, n=10):
    for _ in range(3): ctx.render(frame)
    ctx.kernel_timing(True); ctx.read_kernel_times(); ctx.read_walk_work()
    for _ in range(n): fb, st = ctx.render(frame)
    kt = ctx.read_kernel_times(); ww = ctx.read_walk_work(); ctx.kernel_timing(False)
    ms = [kt.total_ms[k] / max(1, kt.launches[k]) for k in range(5)]
    return fb.copy(), st, ms, ww, n


for name in names:
    scene = ca.HostScene(os.path.join(A, FILES[name]))
    contexts = []
    cfg = ca.default_config(); cfg.shadow_tree = 0; cfg.closest_tree = 0
    ref = ca.GpuContext(0, cfg); ref.upload(scene.desc, build_on_device=True)
    contexts.append(("reference's tree", ref))
    for b, c in [(b, c) for b in builds for c in closest]:
        cfg = ca.default_config(); cfg.shadow_tree = 2; cfg.shadow_tree_build, cfg.shadow_tree_leaf, cfg.shadow_tree_leaf_fill = b
        cfg.closest_tree = 2 if c else 0
        ctx = ca.GpuContext(0, cfg)
        t0 = time.time(); ctx.upload(scene.desc, build_on_device=True); t1 = time.time()
        info = ctx.scene_info()
        print("%s: shadow tree build %s: %d nodes for %d triangles, stack %d, gamma %.3g, built in %.1f ms (upload %.1f ms), scene %.1f MB" %
              (name, b, info.shadow_tree_nodes, info.n_triangles, info.shadow_tree_pending, info.shadow_tree_gamma, info.shadow_tree_ms,
               (t1 - t0) * 1e3, info.hbm_bytes / 1e6), flush=True)
        contexts.append(("tree %d:%d:%d%s" % (b + (" +primary" if c else "",)), ctx))
    for ss in ((True,) if not single else (True, False)):
        frame = scene.frame(size[0], size[1], supersampling=ss)
        base = None
        for label, ctx in contexts:
            fb, st, ms, ww, n = measure(ctx, frame)
            line = "%s %dx%d ss=%d  %-24s frame %.3f  primary %.3f  gate %.3f  shadow %.3f  shade %.3f ms | shadow walk per launch: %d packets, %d nodes, %d records, %d blocked; rays %d + %d" % (
                name, size[0], size[1], ss, label, sum(ms), ms[1], ms[2], ms[3], ms[4], ww.packets[1] // n, ww.quad_nodes[1] // n,
                ww.triangle_records[1] // n, ww.shadow_rays_blocked // n, st.rays_primary, st.rays_shadow)
            line += " | primary walk: %d nodes, %d records, %d left" % (ww.quad_nodes[0] // n, ww.triangle_records[0] // n, st.rays_primary_left)
            if base is None: base = (fb, st)
            else:
                diff = int((fb != base[0]).sum())
                same_rays = (st.rays_primary, st.rays_shadow, st.rays_shadow_elided) == (base[1].rays_primary, base[1].rays_shadow, base[1].rays_shadow_elided)
                line += "  | differing pixels %d, ray counts %s" % (diff, "equal" if same_rays else "DIFFER")
                bad += (diff != 0) + (not same_rays)
            print(line, flush=True)
    for _, ctx in contexts: ctx.close()
print("shadow_tree_check:", "ok" if not bad else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
