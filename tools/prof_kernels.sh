#!/bin/bash
# usage: tools/prof_kernels.sh tag [ENV=..]...   -> per-kernel stats of quick_perf dragon under rocprofv3
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kp_$tag -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py ${SCENES:-dragon} > $GRAFT_REPO_ROOT/gpurun_out/kp_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/kp_$tag "$tag $*"
