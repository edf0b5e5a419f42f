#!/usr/bin/env python3
"""Freezes FNV-1a-64 hashes of ORACLE framebuffers (tests/golden/oracle_regression.json).
These are regression goldens of this repository's oracle -- NOT reference outputs; the
reference pins are in reference_kat.json / color_vectors.json."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import cobbletrace_amd as ca
from oracle import oracle

ASSETS = os.path.join(HERE, "..", "..", "assets")
CASES = [
    # scene, W, H, T, supersampling, subsampling, rng_mode, seed
    ("cube", 640, 480, 8, False, False, 0, 0), ("cube", 320, 240, 8, True, False, 0, 3),
    ("cube", 160, 120, 1, True, False, 1, 1), ("import", 480, 270, 32, False, False, 0, 0),
    ("import", 320, 240, 8, False, True, 0, 0), ("import", 240, 135, 5, True, True, 0, 5),
    ("bunny", 640, 360, 32, False, False, 0, 0), ("bunny", 320, 180, 32, True, False, 0, 0),
    ("bunny", 200, 160, 1, True, False, 1, 1),
]
FILES = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json"}
out = []
for name, w, h, t, ss, sub, rng, seed in CASES:
    scene = ca.HostScene(os.path.join(ASSETS, FILES[name]))
    frame = scene.frame(w, h, threads=t, supersampling=ss, subsampling=sub, rng_mode=rng, rng_seed=seed)
    fb, st = oracle.render(scene.desc, frame, os_threads=8)
    out.append({"scene": name, "width": w, "height": h, "threads": t, "supersampling": ss, "subsampling": sub,
                "rng_mode": rng, "rng_seed": seed, "fnv1a64": "%016x" % oracle.fnv1a64(fb), "rays": st.rays})
    print(out[-1])
json.dump(out, open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=1)
