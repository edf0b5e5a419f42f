"""-m "not gpu": DESIGN appendix D, points 1-2, against the oracle's IntersectTriangle (bvh.cpp:147-163 restated in
oracle/ct_oracle.c) -- no GPU involved.  The cull nodes of the HIP walk leave a triangle test out when the ray's line misses
the triangle's box GROWN by 1/256 of its extent, or meets it only behind the origin, or only beyond the closest hit so far;
the claim is that the reference's own mixed-precision test then returns false or leaves ray.t alone.  Here the claim is put to
the reference's arithmetic directly: one-triangle scenes of many shapes and scales (inside the gate's magnitudes), rays built in
binary64 to pass just OUTSIDE the grown box -- past its faces, edges and corners, by margins from 1e-12 to 1 of its size --
or to have it wholly behind / beyond, and the oracle must report no hit (ray.t unchanged)."""
import numpy as np
import pytest

import cobbletrace_amd as ca
from oracle import oracle

GROW = 1.0 / 256.0


def _triangles(rng, n):
    """shapes from needles to fat triangles, flat on an axis now and then, sizes 1e-3 .. 10, positions within +-20"""
    out = []
    for i in range(n):
        size = 10.0 ** rng.uniform(-3, 1)
        p1 = rng.normal(size=3) * rng.choice([0.0, 1.0, 20.0])
        e1, e2 = rng.normal(size=3) * size, rng.normal(size=3) * size
        if i % 5 == 0:
            e2 = e1 * rng.uniform(-2, 2) + rng.normal(size=3) * size * 1e-3        # a needle
        if i % 7 == 0:
            axis = int(rng.integers(3)); e1[axis] = 0.0; e2[axis] = 0.0           # flat on an axis
        out.append(np.concatenate([p1, p1 + e1, p1 + e2]))
    return np.array(out)


def _grown_box(tri):
    v = tri.reshape(3, 3)
    lo, hi = v.min(0), v.max(0)
    pad = (hi - lo) * GROW
    return lo - pad, hi + pad


def _line_box_interval(o, d, lo, hi):
    """[t_in, t_out] of the line o + t d inside the box, per ray, in binary64 (numpy broadcasting; d has no zero component)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (lo - o) / d, (hi - o) / d
    return np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)


def _rays_around(rng, lo, hi, n):
    """lines through a point just outside the grown box (beyond a face by a relative margin m), running roughly along it"""
    size = max((hi - lo).max(), 1e-12)
    rays = np.zeros(n, dtype=ca.bindings.RAY_DTYPE)
    m = 10.0 ** rng.uniform(-12, 0, n)
    axis = rng.integers(0, 3, n)
    side = rng.integers(0, 2, n)
    point = lo + rng.random((n, 3)) * (hi - lo)
    # push the point out through one face, sometimes through two or three (edges, corners)
    for k in range(3):
        pick = (axis == k) | (rng.random(n) < 0.25)
        point[pick, k] = np.where(side[pick] == 1, hi[k] + m[pick] * size, lo[k] - m[pick] * size)
    d = rng.normal(size=(n, 3))
    d[np.arange(n), axis] *= 10.0 ** rng.uniform(-6, 0, n)             # nearly parallel to that face: stays outside
    d *= (10.0 ** rng.uniform(-2, 2, n))[:, None]
    back = rng.uniform(0.1, 50.0, n)[:, None]
    rays["origin"] = point - d * back
    rays["direction"] = d
    rays["t"] = np.float32(1e30)
    return rays


def test_a_ray_that_misses_the_grown_box_is_rejected_by_the_reference_arithmetic():
    rng = np.random.default_rng(2024)
    checked = behind = beyond = 0
    for tri in _triangles(rng, 160):
        lo, hi = _grown_box(tri)
        scene = ca.HostScene(vertices=tri.reshape(1, 9))
        rays = _rays_around(rng, lo, hi, 6000)
        o, d = rays["origin"], rays["direction"]
        t_in, t_out = _line_box_interval(o, d, lo, hi)
        size = max((hi - lo).max(), 1e-12)
        # (a) the LINE misses the grown box -- with room for the binary64 roundings of this very computation
        scale = np.maximum(np.abs(t_in), np.abs(t_out))
        miss = t_out < t_in - 1e-9 * np.maximum(scale, 1e-300)
        # (b) the box lies wholly behind the origin
        is_behind = ~miss & (t_out < -1e-9 * scale)
        hits = oracle.trace_rays(scene.desc, rays)
        assert (hits["found"][miss] == 0).all(), (tri, int(hits["found"][miss].sum()))
        assert (hits["found"][is_behind] == 0).all(), (tri, "behind", int(hits["found"][is_behind].sum()))
        checked += int(miss.sum()); behind += int(is_behind.sum())
        # (c) wholly beyond a closest hit so far: ray.t stays what it was (tau of appendix D.3 is far below these values)
        ahead = ~miss & (t_in > 0)
        if ahead.any():
            far = rays[ahead].copy()
            start = (t_in[ahead] * (1.0 - 2.0 ** -19)).astype(np.float32)            # the box begins beyond start (1 + 0.93 * 2^-20)
            ok = (t_in[ahead] > start.astype(np.float64) * (1.0 + 2.0 ** -20)) & (start > 1e-2)
            far["t"] = start
            res = oracle.trace_rays(scene.desc, far)
            assert (res["t"][ok].view(np.uint32) == start[ok].view(np.uint32)).all(), (tri, "beyond")
            beyond += int(ok.sum())
        scene.close()
        assert size > 0
    print("rays that miss the grown box: %d, box behind: %d, box beyond ray.t: %d -- none changed ray.t" % (checked, behind, beyond))
    assert checked > 300_000 and behind > 1_000 and beyond > 50_000
