#!/bin/bash
# usage: tools/pmc_run.sh tag "COUNTER LIST" [ENV=..]...
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py ${SCENES:-dragon} > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
