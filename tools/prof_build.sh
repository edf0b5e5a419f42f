#!/bin/bash
# per-kernel times of the device BVH build (ctgpu_build_bvh) under rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kp_build
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kp_build -- python3 $R/tools/time_device_build.py ${1:-dragon} > $R/gpurun_out/kp_build.log 2>&1
cd $R
tail -2 gpurun_out/kp_build.log
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kp_build/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-40s calls %5s total %9.3f ms avg %8.1f us max %9.1f us" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
