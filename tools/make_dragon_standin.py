#!/usr/bin/env python3
"""Deterministic stand-in for models/dragon.ply (the Stanford dragon is not shipped:
/root/reference/.MISSING_LARGE_BLOBS).

A closed, bumpy torus on a periodic 660 x 660 grid -> 435,600 vertices, 871,200
triangles (the real dragon_vrip.ply has 871,414), sized like the real mesh before the
scene's x90 scale: x in +-0.10, y in [0.05, 0.20], z in +-0.05.  Written as ASCII PLY with
6 significant digits and "3 i j k" faces so that it goes through the same parser path
as the bunny (TextCursor::number has no 'e+' support; no value here needs one).

Bit-reproducible on any machine: only IEEE +, -, *, / and rint are used (sin/cos are
an explicit Taylor polynomial, not libm), and decimal formatting is Python's
correctly rounded repr machinery.
"""
import hashlib
import os
import sys

import numpy as np

GRID = 660
TWO_PI = 6.283185307179586


def det_sin(x):
    """sin(x) from basic arithmetic only (deterministic across CPUs / libm builds)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.rint(x / TWO_PI)
    r = x - k * TWO_PI                      # [-pi, pi]
    r2 = r * r
    # Horner form of sum_{n=0}^{15} (-1)^n r^(2n+1)/(2n+1)!
    acc = np.zeros_like(r)
    for n in range(15, 0, -1):
        acc = (acc + 1.0) * (-(r2) / float((2 * n) * (2 * n + 1)))
    return r * (acc + 1.0)


def det_cos(x):
    return det_sin(np.asarray(x, dtype=np.float64) + TWO_PI / 4.0)


def build():
    i = np.arange(GRID, dtype=np.float64)
    u0 = TWO_PI * i / GRID
    # non-uniform spacing along both directions -> a spread of triangle sizes
    u = (u0 + 0.35 * det_sin(u0) + 0.12 * det_sin(3.0 * u0))[:, None]
    v = (u0 + 0.25 * det_sin(2.0 * u0))[None, :]
    bumps = (0.18 * det_sin(5.0 * u + 2.0 * v) + 0.10 * det_sin(11.0 * u - 7.0 * v) +
             0.06 * det_sin(23.0 * u + 17.0 * v) + 0.03 * det_sin(61.0 * u - 43.0 * v))
    major, tube, centre_y = 0.062, 0.024, 0.125
    r = tube * (1.0 + bumps)
    ring = major + r * det_cos(v)
    x = ring * det_cos(u)
    y = centre_y + ring * det_sin(u)
    z = 1.6 * r * det_sin(v) + 0.012 * det_sin(2.0 * u)      # a gentle warp out of the plane
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3)

    ii, jj = np.meshgrid(np.arange(GRID), np.arange(GRID), indexing="ij")
    a = ii * GRID + jj
    b = ((ii + 1) % GRID) * GRID + jj
    c = ((ii + 1) % GRID) * GRID + (jj + 1) % GRID
    d = ii * GRID + (jj + 1) % GRID
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return verts, faces


def write_ply(path, verts, faces):
    lines = ["ply", "format ascii 1.0", "comment CobbleTrace MI355X build: deterministic dragon stand-in",
             "element vertex %d" % len(verts), "property float32 x", "property float32 y", "property float32 z",
             "element face %d" % len(faces), "property list uint8 int32 vertex_indices", "end_header"]
    out = ["\n".join(lines), "\n"]
    out.append("".join("%.6g %.6g %.6g \n" % (p[0], p[1], p[2]) for p in verts.tolist()))
    out.append("".join("3 %d %d %d \n" % (f[0], f[1], f[2]) for f in faces.tolist()))
    text = "".join(out)
    assert "e+" not in text and "nan" not in text and "inf" not in text
    tmp = path + ".tmp"
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)
    return hashlib.sha256(text.encode()).hexdigest()


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "assets", "models", "dragon.ply")
    verts, faces = build()
    digest = write_ply(path, verts, faces)
    print("%s: %d vertices, %d faces, sha256 %s" % (path, len(verts), len(faces), digest))
    print("bounds", verts.min(0), verts.max(0))


if __name__ == "__main__":
    main()
