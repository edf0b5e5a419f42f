#!/usr/bin/env python3
"""Generates tests/golden/color_vectors.json by CALLING the reference's own color.h / mymath.h
(compiled in place into oracle/_ref/libref_headers.so by `make -C oracle ref`).  Run in the
build container, where /root/reference exists; the JSON holds inputs and expected outputs only."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libref_headers.so"))
lib.ref_shade_value.restype = C.c_uint32
lib.ref_shade_value.argtypes = [C.c_uint32, C.c_float]
lib.ref_blend_sample.restype = C.c_uint32
lib.ref_blend_sample.argtypes = [C.c_uint32, C.c_uint32]
lib.ref_mix_reflection.restype = C.c_uint32
lib.ref_mix_reflection.argtypes = [C.c_uint32, C.c_uint32, C.c_float]

rng = np.random.default_rng(2026)
colors = [0x000000, 0xffffff, 0x0000ff, 0x00ff00, 0xff0000, 0xff9bff, 0x66e6ff, 0x66ffcc, 0x333333, 0x010203]
colors += [int(c) for c in rng.integers(0, 1 << 24, 90)]
values = [0.0, 0.2, 0.5, 0.8, 1.0, 1.0000001, 1.7, 3.0, 1e-7] + [float(np.float32(v)) for v in rng.random(16) * 1.3]
out = {"shade_value": [], "blend_sample": [], "mix_reflection": []}
for c in colors:
    for v in values:
        out["shade_value"].append(["%06x" % c, float(np.float32(v)), "%06x" % lib.ref_shade_value(c, v)])
for a in colors[:40]:
    for b in colors[20:45]:
        out["blend_sample"].append(["%06x" % a, "%06x" % b, "%06x" % lib.ref_blend_sample(a, b)])
        out["mix_reflection"].append(["%06x" % a, "%06x" % b, 0.5, "%06x" % lib.ref_mix_reflection(a, b, 0.5)])
        out["mix_reflection"].append(["%06x" % a, "%06x" % b, 0.25, "%06x" % lib.ref_mix_reflection(a, b, 0.25)])
json.dump(out, open(os.path.join(HERE, "color_vectors.json"), "w"))
print({k: len(v) for k, v in out.items()})
