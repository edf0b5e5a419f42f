#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh into profiles-ready summaries:
  TAG_summary.md    per-kernel average duration and counters
  TAG_traffic.json  HBM bytes per launch of every kernel (FETCH_SIZE doubled, as MI355X_MICROARCH.md's HBM
                    section prescribes for gfx950, + WRITE_SIZE; separate --pmc passes)
  TAG_issue.json    instruction-issue figures of the two walk kernels (what actually bounds them)
Each JSON carries the hash of the kernel sources it was measured on; bench.py quotes it only when that hash
is the one on disk."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                        # kernel names and the source hash live there

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = "gpurun_out"
MATCH = (("ct_sample_grid", "ct_sample_grid"), ("ct_walk_closest_tree", "ct_walk_closest_tree"), ("ct_walk_packet<0", "ct_walk_packet<primary>"), ("ct_walk_shadow_tree", "ct_walk_shadow_tree"),
         ("ct_light_gate", "ct_light_gate"), ("ct_walk_packet<1", "ct_walk_packet<shadow>"), ("ct_shade", "ct_shade"),
         ("ct_assemble", "ct_assemble"))
CLOCK_HZ, SIMDS = 2.4e9, 1024                      # MI355X: 256 CUs x 4 SIMDs


def short(name):
    for needle, label in MATCH:
        if needle in name:
            return label
    return None


def newest(pattern):
    found = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return found[-1:]                               # gpurun_out accumulates earlier collections: keep the latest


stats = {}
for f in newest("%s/%s_stats/**/*kernel_stats.csv" % (out, tag)):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("fetch", "write", "cache", "sq", "sq2", "sqc"):
    for f in newest("%s/%s_%s/**/*counter_collection.csv" % (out, tag, part)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else 0.0

lines = ["# rocprofv3 summary %s -- `python bench.py` (dragon stand-in 1920x1080, 16 spp, 1 MI355X)" % tag, "",
         "| kernel | launches | avg ms | min | max | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes / launch | L2 hit % | scalar cache hit % |",
         "|---|---|---|---|---|---|---|---|---|---|"]
frame_ms = 0.0
per_launch = {}
for _, k in MATCH:
    if k not in stats:
        continue
    c = counters[k]
    hit, miss = mean(c["TCC_HIT_sum"]), mean(c["TCC_MISS_sum"])
    sq_hit, sq_req = mean(c["SQC_DCACHE_HITS"]), mean(c["SQC_DCACHE_REQ"])
    hbm = int((2 * mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"])) * 1024)
    per_launch[k] = hbm
    lines.append("| %s | %d | %.3f | %.3f | %.3f | %.0f | %.0f | %.3g | %.1f | %.1f |" % (
        k, stats[k][0], stats[k][1], stats[k][2], stats[k][3], mean(c["FETCH_SIZE"]), mean(c["WRITE_SIZE"]), hbm,
        100 * hit / max(hit + miss, 1), 100 * sq_hit / max(sq_req, 1)))
    frame_ms += stats[k][1]
lines += ["", "frame (sum of the launches above): %.3f ms; HBM traffic per frame %.2f GB" % (frame_ms, sum(per_launch.values()) / 1e9), ""]

issue = {"kernel_source_hash": bench.kernel_source_hash(), "tag": tag, "unit_note":
         "SQ_* cycle counters are in 4-clock issue quanta summed over waves; utilisation = quanta x 4 / (1024 SIMDs x kernel clocks at 2.4 GHz)",
         "kernels": {}}
lines += ["## Instruction issue of the walk kernels (rocprofv3 --pmc, passes of their own)", "",
          "| kernel | VALU instr | SALU instr | VALU pipe busy % | SALU / CU-cycle % | lane util % | wave: issuing % | waiting for issue % | in s_waitcnt % |",
          "|---|---|---|---|---|---|---|---|---|"]
for k in ("ct_walk_closest_tree", "ct_walk_packet<primary>", "ct_walk_shadow_tree", "ct_walk_packet<shadow>"):
    if k not in stats or not counters[k]["SQ_WAVE_CYCLES"]:
        continue
    c = counters[k]
    clocks = stats[k][1] * 1e-3 * CLOCK_HZ
    valu_busy = 100.0 * mean(c["SQ_ACTIVE_INST_VALU"]) * 4 / (SIMDS * clocks)
    salu_rate = 100.0 * mean(c["SQ_INSTS_SALU"]) / (256 * clocks)            # one scalar unit per CU, one instruction per clock
    lane = 100.0 * mean(c["SQ_THREAD_CYCLES_VALU"]) / max(mean(c["SQ_ACTIVE_INST_VALU"]) * 64, 1)
    wc = mean(c["SQ_WAVE_CYCLES"])
    rec = {"avg_ms": round(stats[k][1], 4), "valu_instructions": int(mean(c["SQ_INSTS_VALU"])), "salu_instructions": int(mean(c["SQ_INSTS_SALU"])),
           "valu_pipe_busy_frac": round(valu_busy / 100, 3), "salu_per_cu_cycle_frac": round(salu_rate / 100, 3),
           "valu_lane_utilisation": round(lane / 100, 3),
           "wave_cycles_issuing_frac": round(mean(c["SQ_ACTIVE_INST_ANY"]) / wc, 3) if c["SQ_ACTIVE_INST_ANY"] else None,
           "wave_cycles_waiting_for_issue_frac": round(mean(c["SQ_WAIT_INST_ANY"]) / wc, 3) if c["SQ_WAIT_INST_ANY"] else None,
           "wave_cycles_in_waitcnt_frac": round(mean(c["SQ_WAIT_ANY"]) / wc, 3)}
    issue["kernels"][k] = rec
    pct = lambda v: "-" if v is None else "%.0f" % (100 * v)
    lines.append("| %s | %.3g | %.3g | %.0f | %.0f | %.0f | %s | %s | %s |" % (
        k, rec["valu_instructions"], rec["salu_instructions"], valu_busy, salu_rate, lane, pct(rec["wave_cycles_issuing_frac"]),
        pct(rec["wave_cycles_waiting_for_issue_frac"]), pct(rec["wave_cycles_in_waitcnt_frac"])))
lines.append("")
for f in glob.glob("%s/%s_stats.log" % (out, tag)):
    for l in open(f):
        if l.startswith("{"):
            lines += ["bench line of the profiled run (`--steps 10 --warmup 2 --no-cpu-baseline --inflight 1`):", "", "```", l.strip(), "```"]
open("%s/%s_summary.md" % (out, tag), "w").write("\n".join(lines) + "\n")
json.dump({"tag": tag, "kernel_source_hash": bench.kernel_source_hash(), "frame_ms_sum_of_kernels": frame_ms,
           "hbm_bytes_per_launch": per_launch, "hbm_bytes_per_frame": sum(per_launch.values()),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "(gfx950: FETCH_SIZE tallies 128-byte requests at 64 B, MI355X_MICROARCH.md HBM section)",
           "workload": "scene_import_dragon.json 1920x1080 T=32 supersampling, python bench.py --inflight 1"},
          open("%s/%s_traffic.json" % (out, tag), "w"), indent=1)
json.dump(issue, open("%s/%s_issue.json" % (out, tag), "w"), indent=1)
print("\n".join(lines))
