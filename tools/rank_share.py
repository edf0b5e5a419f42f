#!/usr/bin/env python3
"""What ONE rank of an N-GPU job does per frame, rehearsed on one GPU: trace rank 0's row tiles of the headline frame,
assemble a whole frame's worth of tiles (its own, repeated: the gather's result has that size) and copy the frame to a
pinned host buffer -- with F frames in flight and the library's grid / chunk knobs.  The estimate it gives for the
N-GPU frame rate is rank 0's time per frame (the other ranks do less); the collective itself is not in it.
    python tools/rank_share.py [scene]      env: WORLDS=2,4,8  INFLIGHT=3,4,6  BLOCKS=0,12  CHUNKS=0,128"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cobbletrace_amd as ca
import __graft_entry__ as entry

entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
frame = scene.frame(1920, 1080, rng_mode=ca.RNG_HASH, rng_seed=0)
ints = lambda key, default: [int(v) for v in os.environ.get(key, default).split(",")]
for world in ints("WORLDS", "1,2,4,8"):
    for blocks in ints("BLOCKS", "0,12"):
        for chunk in ints("CHUNKS", "0,128"):
            for inflight in ints("INFLIGHT", "3,4,6"):
                os.environ["CTGPU_BLOCKS_PER_CU"], os.environ["CTGPU_CHUNK"] = str(blocks), str(chunk)
                part = ca.Partition(8, 0, world)
                slots = []
                for i in range(inflight):
                    ctx = ca.GpuContext(0)
                    if i == 0:
                        ctx.upload(scene.desc, build_on_device=True)
                    else:
                        ctx.share_scene(slots[0][0])
                    n = ctx.partition_pixels(frame, part)
                    slots.append((ctx, torch.cuda.Stream(), torch.zeros(n * world, dtype=torch.int32, device="cuda"),
                                  torch.zeros(frame.height * frame.width, dtype=torch.int32, device="cuda"),
                                  torch.zeros(frame.height * frame.width, dtype=torch.int32).pin_memory()))

                def one(i):
                    ctx, stream, tiles, fb, host = slots[i % inflight]
                    with torch.cuda.stream(stream):
                        ctx.render_partition(frame, part, tiles.data_ptr(), stream.cuda_stream)
                        ctx.assemble(frame, part, tiles.data_ptr(), fb.data_ptr(), stream.cuda_stream)
                        ctx.download_frame(frame, fb.data_ptr(), host.data_ptr(), stream.cuda_stream)

                for i in range(2 * inflight):
                    one(i)
                torch.cuda.synchronize()
                n_frames = int(os.environ.get("FRAMES", "60"))
                t0 = time.perf_counter()
                for i in range(n_frames):
                    one(i)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n_frames * 1e3
                st = slots[0][0].read_counters()
                print("%s world %d blocks/CU %2d chunk %3d inflight %d: %.3f ms per frame on rank 0  (walks: %s)" % (
                    name, world, blocks, chunk, inflight, ms, ", ".join(k for k in ("ordered", "cull_tree", "shadow_tree", "closest_tree") if st.path()[k]) or "reference's tree"), flush=True)
                for s in slots:
                    s[0].close()
