#!/bin/bash
# tools/pmc_compare.sh OTHER_LIBRARY -- on the GPU box: the walks' instruction counters (one rocprofv3 --pmc pass each) for
# the library in the tree and for another build of it, side by side.  Output: gpurun_out/pmc_compare.txt
other=$(realpath "$1")
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for which in current other; do
  rm -rf $R/gpurun_out/pmc_$which
  if [ $which = other ]; then export CTGPU_LIBRARY=$other CTGPU_ALLOW_FOREIGN_LIBRARY=1 CTGPU_OLDER_LIBRARY=1; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_$which -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/pmc_$which.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$which.log; exit 1; }
done
python3 - <<PY > $R/gpurun_out/pmc_compare.txt
import csv, glob, collections
for which in ("other", "current"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob("$R/gpurun_out/pmc_%s/**/*counter_collection.csv" % which, recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "ct_walk_packet" not in name: continue
            kind = "shadow" if "ct_walk_packet<1" in name or "ILi1E" in name else "primary"
            a = acc[(kind, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for (kind, counter), (total, n) in sorted(acc.items()):
        print("%-8s %-8s %-22s %14.0f per launch (%d launches)" % (which, kind, counter, total / n, n))
PY
cat $R/gpurun_out/pmc_compare.txt
