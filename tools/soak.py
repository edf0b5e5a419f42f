#!/usr/bin/env python3
"""A long mixed session in ONE process: scenes, sizes, sampling modes and entry points in random order -- ctgpu_render into
pageable and page-locked memory, begin / poll, render_multi over three contexts of one device, contexts created and closed,
uploads with and without a tree -- every frame compared with the first frame of its kind (rendered once by a fresh context).
     python tools/soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
files = {"cube": "scene_file_cube.json", "import": "scene_import.json", "bunny": "scene_import_bunny.json", "dragon": "scene_import_dragon.json"}
scenes = {k: ca.HostScene(os.path.join(A, v)) for k, v in files.items()}
sizes = [(320, 240), (640, 360), (641, 479), (960, 540), (1920, 1080)]
reference, frames, t_end, t_print = {}, 0, time.time() + budget, time.time() + 30
ctxs = {}
while time.time() < t_end:
    name = str(rng.choice(list(files)))
    w, h = sizes[int(rng.integers(len(sizes)))]
    if name == "cube" and w > 700: w, h = 640, 480
    ss = bool(rng.integers(2))
    frame = scenes[name].frame(w, h, supersampling=ss, rng_seed=int(rng.integers(3)))
    key = (name, w, h, ss, frame.rng_seed)
    if key not in reference:
        fresh = ca.GpuContext(0); fresh.upload(scenes[name].desc); reference[key] = fresh.render(frame)[0].copy(); fresh.close()
    if name not in ctxs or rng.random() < 0.05:
        if name in ctxs:
            for c in ctxs.pop(name): c.close()
        first = ca.GpuContext(0); first.upload(scenes[name].desc, build_on_device=bool(rng.integers(2)))
        others = [ca.GpuContext(0) for _ in range(2)]
        for c in others: c.share_scene(first)
        ctxs[name] = [first] + others
    group = ctxs[name]
    how = int(rng.integers(5))
    if how == 0:
        fb, _ = group[0].render(frame)                                        # a fresh pageable array
    elif how == 1:
        pinned = torch.zeros(h, w, dtype=torch.int32).pin_memory()
        fb, _ = group[1].render(frame, pinned.numpy().view(np.uint32))
    elif how == 2:
        fb = group[2].render_begin(frame)
        while not group[2].render_poll()[0]: pass
    elif how == 3:
        fb, _ = ca.GpuContext.render_multi(group, frame)
    else:
        keep = np.zeros((h, w), dtype=np.uint32)
        group[0].keep_framebuffer_pinned(True)
        group[0].render_begin(frame, keep); group[0].render_poll(wait=True)
        group[0].keep_framebuffer_pinned(False)
        fb = keep
    if not (fb == reference[key]).all():
        print("MISMATCH", key, how, int((fb != reference[key]).sum())); sys.exit(1)
    frames += 1
    if time.time() > t_print:
        print('%d frames, %d kinds' % (frames, len(reference)), flush=True); t_print = time.time() + 30
print("%d frames of %d kinds through five entry points: every one equal to its reference frame" % (frames, len(reference)))
