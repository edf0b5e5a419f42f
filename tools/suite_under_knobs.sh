#!/bin/bash
# tools/suite_under_knobs.sh TAG -- the -m gpu suite once per knob that selects another code path (tests that examine a knob
# themselves set it themselves); one log per run under gpurun_out/.
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
cd $R
run() {
  name=$1; shift
  env "$@" timeout -k 10 600 python3 -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests_$name.log 2>&1
  echo "$name ($*): $(tail -1 gpurun_out/${tag}_gpu_tests_$name.log)"
}
run packet0 CTGPU_PACKET=0
run keep CTGPU_KEEP_SHADOW_RAYS=1
run bounce0 CTGPU_BOUNCE=0
run packhost CTGPU_PACK=host
run order0 CTGPU_ORDER=0
run bigleaf0 CTGPU_BIG_LEAF=0
run bigleaf4all CTGPU_BIG_LEAF=4 CTGPU_BIG_LEAF_SHARE=0
run split CTGPU_SHADOW_SPLIT=0.01
