#!/usr/bin/env python3
"""Two builds of libctgpu.so beside each other on the same box: per-walk times of the headline frame, alternating.
     python tools/ab_library.py variants/libctgpu_r3.so [more.so ...] [scene] [reps]
(the other library may be an earlier round's: symbols it lacks are skipped, CTGPU_OLDER_LIBRARY)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("AB_CHILD") == "1":
    sys.path.insert(0, os.path.join(HERE, ".."))
    import cobbletrace_amd as ca
    import __graft_entry__ as entry
    entry.prepare_assets(need_dragon=True)
    name = sys.argv[1]
    files = {"import": "scene_import.json", "bunny": "scene_import_bunny.json", "dragon": "scene_import_dragon.json"}
    scene = ca.HostScene(os.path.join(HERE, "..", "assets", files[name]))
    ctx = ca.GpuContext(0)
    ctx.upload(scene.desc, build_on_device=True)
    for ss in (True, False):
        frame = scene.frame(1920, 1080, supersampling=ss)
        for _ in range(4): ctx.render(frame)
        ctx.kernel_timing(True); ctx.read_kernel_times()
        n = 12
        for _ in range(n): fb, st = ctx.render(frame)
        kt = ctx.read_kernel_times(); ctx.kernel_timing(False)
        ms = [kt.total_ms[k] / max(1, kt.launches[k]) for k in range(5)]
        print("RESULT ss=%d frame %.3f primary %.3f shadow %.3f gate %.3f shade %.3f" % (ss, sum(ms), ms[1], ms[3], ms[2], ms[4]))
    ctx.close()
    sys.exit(0)
others = [os.path.abspath(a) for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
name = rest[0] if rest else "dragon"
reps = int(rest[1]) if len(rest) > 1 else 3
for r in range(reps):
    for label, lib in [(os.path.basename(o)[:-3].replace("libctgpu_", "").ljust(10), o) for o in others] + [("current".ljust(10), None)]:
        env = dict(os.environ, AB_CHILD="1")
        if lib: env.update(CTGPU_LIBRARY=lib, CTGPU_OLDER_LIBRARY="1", CTGPU_ALLOW_FOREIGN_LIBRARY="1")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("RESULT"): print("%s %s %s" % (name, label, line[7:]), flush=True)
        if out.returncode != 0: print(out.stderr[-1500:])
