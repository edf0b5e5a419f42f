#!/bin/bash
# usage: tools/prof_kernels.sh tag [ENV=..]...   -> per-kernel stats of quick_perf dragon under rocprofv3
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kp_$tag -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py dragon > $GRAFT_REPO_ROOT/gpurun_out/kp_$tag.log 2>&1
echo "== $tag $*"
cat $GRAFT_REPO_ROOT/gpurun_out/kp_$tag/*/*kernel_stats.csv | grep -E "ct_" | awk -F'","' '{printf "%-70s calls %s avg %.3f ms\n", substr($1,2,68), $2, $4/1e6}'
