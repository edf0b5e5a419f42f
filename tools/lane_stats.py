#!/usr/bin/env python3
"""How many of a packet's 64 rays take part in what the shared walk does (the CT_WALK_STATS diagnostic build):
histograms of node visits, triangle tests and packets by the number of rays taking part, per walk.
    make -C cobbletrace_amd/csrc libctgpu_stats.so && python tools/lane_stats.py [scene ...]     (SS=0: single-sample frames)
A node visit costs the wave the same whatever the number of rays in it (every lane computes the four boxes), a triangle
test likewise (the other lanes idle): "useful" = rays taking part / 64, weighted by how often."""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("LANE_STATS_CHILD") != "1":
    # the counters come out on the library's stderr: run the measurement in a child and read it
    names = sys.argv[1:] or ["dragon"]
    for name in names:
        for kind, label in ((0, "primary"), (1, "shadow")):
            env = dict(os.environ, LANE_STATS_CHILD="1", CTGPU_LIBRARY=os.path.join(HERE, "..", "cobbletrace_amd", "csrc", "libctgpu_stats.so"),
                       CTGPU_DEBUG_PACKET="1", CTGPU_WALK_THRESH="64,%d,24" % kind, CTGPU_ORDER="0")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                sys.exit(out.stderr[-2000:])
            hists = {}
            for line in out.stderr.splitlines():
                m = re.match(r"hist (.+?) by rays taking part.*?:((?: \d+)+)", line)
                if m:
                    hists[m.group(1)] = [int(v) for v in m.group(2).split()]          # (the last frame's)
            print("%s, %s walk (%s)" % (name, label, "16 spp" if os.environ.get("SS", "1") != "0" else "1 spp"))
            for what in ("packets", "node visits", "triangle tests"):
                h = hists[what]
                total = sum(h)
                mid = [8 * b + 4.5 for b in range(8)]
                useful = sum(n * m for n, m in zip(h, mid)) / max(total, 1) / 64
                print("   %-15s %11d   share by rays taking part 1-8 .. 57-64: %s   mean %.0f %% of the wave" %
                      (what, total, " ".join("%4.1f" % (100.0 * n / max(total, 1)) for n in h), 100 * useful))
    sys.exit(0)
sys.path.insert(0, os.path.join(HERE, ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
name = sys.argv[1]
files = {"import": "scene_import.json", "cube": "scene_file_cube.json"}
scene = ca.HostScene(os.path.join(HERE, "..", "assets", files.get(name, "scene_import_%s.json" % name)))
ctx = ca.GpuContext(0)
ctx.upload(scene.desc, build_on_device=True)
frame = scene.frame(int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), supersampling=os.environ.get("SS", "1") != "0")
ctx.render(frame)
ctx.close()
