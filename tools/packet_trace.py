#!/usr/bin/env python3
"""Timeline of the packet walks of one frame, from the diagnostic build of the library:
     make -C cobbletrace_amd/csrc libctgpu_trace.so
     CTGPU_LIBRARY=$PWD/cobbletrace_amd/csrc/libctgpu_trace.so python tools/packet_trace.py [scene ...]   (env SS=0/1, W, H)
   (the trace itself costs one atomic per packet: fine for single-sample frames, it doubles a supersampled one)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cobbletrace_amd as ca
import __graft_entry__ as entry
entry.prepare_assets(need_dragon=True)
A = os.path.join(os.path.dirname(__file__), "..", "assets")
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
ss = os.environ.get("SS", "0") != "0"
lib = C.CDLL(os.environ["CTGPU_LIBRARY"])
def fetch():
    buf = np.zeros((1 << 20, 5), dtype=np.uint64); n = C.c_uint()
    assert lib.ctgpu_debug_packet_trace(C.c_void_p(buf.ctypes.data), 1 << 20, C.byref(n)) == 0
    return buf[:n.value]
for name in (sys.argv[1:] or ["dragon"]):
    scene = ca.HostScene(os.path.join(A, "scene_import_%s.json" % name))
    ctx = ca.GpuContext(0); ctx.upload(scene.desc)
    f = scene.frame(W, H, supersampling=ss)
    for _ in range(3): ctx.render(f)
    fetch()
    fb, st = ctx.render(f)
    r = fetch()
    print("%s ss=%d kernel %.3f ms, %d packet records" % (name, ss, st.kernel_ms, len(r)))
    kind = (r[:, 0] >> np.uint64(56)).astype(int); block = ((r[:, 0] >> np.uint64(32)) & np.uint64(0xffffff)).astype(int)
    rays = (r[:, 0] & np.uint64(0xffffffff)).astype(int)
    t0 = r[:, 1].astype(np.int64); t1 = r[:, 2].astype(np.int64)
    quads = (r[:, 3] >> np.uint64(32)).astype(int); recs = (r[:, 3] & np.uint64(0xffffffff)).astype(int)
    for k, label in ((0, "primary"), (1, "shadow")):
        m = kind == k
        if not m.any(): continue
        b, e = t0[m], t1[m]; origin = b.min()
        us = (e - b) / 100.0; span = (e.max() - origin) / 100.0
        v = quads[m] + recs[m]
        print(" %s: %d packets, span %.1f us (first start -> last end); last packet START at %.1f us" % (label, m.sum(), span, (b.max() - origin) / 100.0))
        print("   packet duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f ; sum/wave-slots(%d) = %.1f us" % (
              us.mean(), np.percentile(us, 50), np.percentile(us, 90), np.percentile(us, 99), us.max(), len(set(block[m])), us.sum() / len(set(block[m]))))
        print("   visits/packet: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %d ; us per visit (packets with >=20 visits): %.2f" % (
              v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), (us[v >= 20] / v[v >= 20]).mean() if (v >= 20).any() else 0))
        # how many waves are inside a packet at each tenth of the span
        edges = np.linspace(0, span, 11)
        act = [int(((b - origin) / 100.0 <= x) .__and__((e - origin) / 100.0 > x).sum()) for x in edges[:-1] + span / 20]
        print("   packets in flight at 5%%,15%%..95%% of the span: %s" % act)
        # per wave
        busy = np.zeros(block.max() + 1); cnt = np.zeros(block.max() + 1)
        np.add.at(busy, block[m], us); np.add.at(cnt, block[m], 1)
        used = cnt > 0
        print("   per wave: packets mean %.1f max %d ; busy us mean %.1f max %.1f" % (cnt[used].mean(), cnt.max(), busy[used].mean(), busy.max()))
        top = np.argsort(-us)[:6]
        for i in top:
            print("     long packet: %.1f us, start %.1f us, rays %d, quads %d, records %d, extra %d" % (us[i], (b[i] - origin) / 100.0, rays[m][i], quads[m][i], recs[m][i], int(r[m][i, 4])))
    ctx.close()
