#!/usr/bin/env python3
"""Frames far beyond the benchmarked sizes, to see that nothing in the path is sized for 1080p: the dragon at 7680x4320 with
16 samples per pixel (299 M samples, 0.9 G rays) and at 16384x16384 single-sample (268 M pixels), each rendered twice
(idempotence), ray accounting checked in closed form, two row bands of each against the oracle bit for bit.
     python tools/max_size_check.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cobbletrace_amd as ca
import __graft_entry__ as entry
from oracle import oracle as orc                            # the checker (tools/ are measurement and checking scripts)

entry.prepare_assets(need_dragon=True)
scene = ca.HostScene(os.path.join(os.path.dirname(__file__), "..", "assets", "scene_import_dragon.json"))
ctx = ca.GpuContext(0)
ctx.upload(scene.desc)
O = orc
bad = 0
for (w, h, ss, threads, bands) in [(7680, 4320, True, 135, [60, 75]), (16384, 16384, False, 512, [200, 300])]:
    frame = scene.frame(w, h, threads=threads, supersampling=ss, rng_mode=ca.RNG_HASH, rng_seed=0)
    t0 = time.time(); a, sa = ctx.render(frame); t1 = time.time(); b, sb = ctx.render(frame); t2 = time.time()
    q = h // threads
    rows, cols, spp = threads * q, min(w, h), 16 if ss else 1
    want_primary = (rows * cols - cols) * spp
    same = bool((a == b).all()) and sa.rays == sb.rays
    hits = int(((a != 0x333333) & (a != 0)).sum())
    print("%dx%d %s: %.1f / %.1f ms wall, kernel %.2f ms, %d rays (%d primary, expected %d), %d pixels on the mesh, twice the same: %s"
          % (w, h, "16 spp" if ss else "1 spp", (t1 - t0) * 1e3, (t2 - t1) * 1e3, sb.kernel_ms, sa.rays, sa.rays_primary, want_primary, hits, same), flush=True)
    if sa.rays_primary != want_primary or not same: bad += 1
    for band in bands:
        part, _ = O.render(scene.desc, frame, os_threads=1, band_first=band, band_count=1)
        r = part.any(axis=1)
        n_diff = int((a[r] != part[r]).sum())
        print("   band %d (%d rows) against the oracle: %d differing pixels of %d" % (band, int(r.sum()), n_diff, int(r.sum()) * w), flush=True)
        bad += n_diff != 0
    del a, b
print("max-size check:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
