#!/usr/bin/env python3
"""How far does a WAVE get into IntersectTriangle (bvh.cpp:147-163) per triangle record of the shared walk?  The test's
   early exits are wave-level branches already (a stage costs only when some lane reaches it), so a cheaper binary32
   reject level in front of it can only pay for records on which EVERY lane is out early -- this counts them.
     make -C cobbletrace_amd/csrc libctgpu_tristats.so && python tools/tri_stats.py [scene ...]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["CTGPU_LIBRARY"] = os.path.join(HERE, "..", "cobbletrace_amd", "csrc", "libctgpu_tristats.so")
os.environ["CTGPU_DEBUG_TRI"] = "1"
sys.path.insert(0, os.path.join(HERE, ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
for name in (sys.argv[1:] or ["dragon", "bunny"]):
    scene = ca.HostScene(os.path.join(HERE, "..", "assets", "scene_import.json" if name == "import" else "scene_import_%s.json" % name))
    for ss in (True, False):
        for kind, label in ((0, "primary"), (1, "shadow")):
            os.environ["CTGPU_WALK_THRESH"] = "64,%d,24" % kind         # (the middle number picks the walk the build counts)
            ctx = ca.GpuContext(0)
            ctx.upload(scene.desc, build_on_device=True)
            sys.stderr.write("%s ss=%d %s walk: " % (name, ss, label))
            sys.stderr.flush()
            ctx.render(scene.frame(1920, 1080, supersampling=ss))
            ctx.close()
