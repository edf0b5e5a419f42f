#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh into profiles-ready summaries:
TAG_summary.md (per-kernel average duration, counters) and TAG_traffic.json (HBM bytes per frame,
FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950)."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = "gpurun_out"
FRAME_KERNELS = ("ct_sample_grid", "ct_walk_packet<0", "ct_walk_packet<1", "ct_walk<0", "ct_walk<1", "ct_shade", "ct_assemble")


def short(name):
    for k in FRAME_KERNELS:
        if k in name:
            return k + (">" if "<" in k else "")
    if "ct_trace_tiles" in name:
        return "ct_trace_tiles(count)" if "true>(" in name.replace(" ", "") and name.count("true") == 2 or ", true>" in name else "ct_trace_tiles"
    return None


stats = {}
def newest(pattern):
    found = sorted(glob.glob(pattern), key=os.path.getmtime)
    return found[-1:]                       # gpurun_out accumulates earlier collections: keep the latest


for f in newest("%s/%s_stats/*/*kernel_stats.csv" % (out, tag)):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("fetch", "write", "cache", "sq", "sqc"):
    for f in newest("%s/%s_%s/*/*counter_collection.csv" % (out, tag, part)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else 0.0

ORDER = ("ct_sample_grid", "ct_walk_packet<0>", "ct_walk_packet<1>", "ct_walk<0>", "ct_walk<1>", "ct_shade", "ct_assemble")
launches = {k: 1 for k in ORDER}
lines = ["# rocprofv3 summary %s -- `python bench.py` (dragon stand-in 1920x1080, 16 spp, 1 MI355X)" % tag, "",
         "| kernel | launches/frame | avg ms | FETCH_SIZE KB | WRITE_SIZE KB | L2 hit % | scalar cache hit % | VALU busy % | VALU lane util % | wave wait % |",
         "|---|---|---|---|---|---|---|---|---|---|"]
frame_ms = fetch_kb = write_kb = 0.0
for k in ORDER:
    if k not in stats:
        continue
    c = counters[k]
    n = launches[k]
    hit, miss = mean(c["TCC_HIT_sum"]), mean(c["TCC_MISS_sum"])
    tcp_total, tcp_miss = mean(c["TCP_TOTAL_CACHE_ACCESSES_sum"]), mean(c["TCP_TCC_READ_REQ_sum"])
    util = 100.0 * mean(c["SQ_THREAD_CYCLES_VALU"]) / (mean(c["SQ_ACTIVE_INST_VALU"]) * 64) if c["SQ_ACTIVE_INST_VALU"] else 0
    sq_hit, sq_req = mean(c["SQC_DCACHE_HITS"]), mean(c["SQC_DCACHE_REQ"])
    # SQ_ACTIVE_INST_VALU counts 4-cycle issue slots summed over the 1024 SIMDs (256 CUs x 4); clock from the stats pass
    valu_busy = 100.0 * mean(c["SQ_ACTIVE_INST_VALU"]) * 4 / (1024 * stats[k][1] * 1e-3 * 2.4e9) if c["SQ_ACTIVE_INST_VALU"] else 0
    wait = 100.0 * mean(c["SQ_WAIT_ANY"]) / mean(c["SQ_WAVE_CYCLES"]) if c["SQ_WAVE_CYCLES"] else 0
    lines.append("| %s | %d | %.3f | %.0f | %.0f | %.1f | %.1f | %.0f | %.0f | %.0f |" % (
        k, n, stats[k][1], mean(c["FETCH_SIZE"]), mean(c["WRITE_SIZE"]), 100 * hit / max(hit + miss, 1),
        100 * sq_hit / max(sq_req, 1), valu_busy, util, wait))
    frame_ms += n * stats[k][1]
    fetch_kb += n * mean(c["FETCH_SIZE"])
    write_kb += n * mean(c["WRITE_SIZE"])
hbm_bytes = int((2 * fetch_kb + write_kb) * 1024)
lines += ["", "frame (sum of the launches above): %.3f ms" % frame_ms,
          "HBM traffic per frame: FETCH_SIZE %.0f KB (x2 gfx950 correction) + WRITE_SIZE %.0f KB = %.2f GB" %
          (fetch_kb, write_kb, hbm_bytes / 1e9), ""]
for f in glob.glob("%s/%s_stats.log" % (out, tag)):
    for l in open(f):
        if l.startswith("{"):
            lines += ["bench line of the profiled run:", "", "```", l.strip(), "```"]
open("%s/%s_summary.md" % (out, tag), "w").write("\n".join(lines) + "\n")
json.dump({"tag": tag, "frame_ms_sum_of_kernels": frame_ms, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
           "hbm_bytes_per_frame": hbm_bytes,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
           "workload": "scene_import_dragon.json 1920x1080 T=32 supersampling"},
          open("%s/%s_traffic.json" % (out, tag), "w"), indent=1)
print("\n".join(lines[:14]))
