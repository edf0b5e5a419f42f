#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_icache
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_SMEM SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $R/gpurun_out/pmc_icache -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/pmc_icache.log 2>&1 || { tail -5 $R/gpurun_out/pmc_icache.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("$R/gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if "ct_walk_packet" not in name: continue
        kind = "shadow" if "ct_walk_packet<1" in name else "primary"
        a = acc[(kind, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
for (kind, counter), (total, n) in sorted(acc.items()):
    print("%-8s %-22s %14.0f per launch (%d)" % (kind, counter, total / n, n))
PY
