#!/bin/bash
# tools/args_skew_check.sh -- the shadow walk re-reads its kernel arguments from the argument segment (ct_pipeline.hip,
# kernarg_again) and checks the layout it assumes before using it.  This builds the library with a deliberately wrong
# offset (-DCT_ARGS_SKEW=8) and shows that a frame then ends with an error instead of a picture.  Run on a GPU box.
set -e
cd "$(dirname "$0")/../cobbletrace_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-flush-denormals-to-zero -I../../include -I. -Wno-unused-function -Wno-bitwise-instead-of-logical -DCT_ARGS_SKEW=8 \
    -shared -o ../../variants/libctgpu_skew.so ct_kernels.hip ct_pipeline.hip ctgpu_api.hip ctgpu_comm.hip ct_build.hip ct_pack.hip -ldl
cd ../..
CTGPU_LIBRARY=variants/libctgpu_skew.so python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import cobbletrace_amd as ca
scene = ca.HostScene("assets/scene_import.json")
ctx = ca.GpuContext(0); ctx.upload(scene.desc)
try:
    ctx.render(scene.frame(320, 240, supersampling=True))
    print("FAILED: the skewed library rendered a frame"); sys.exit(1)
except Exception as e:
    print("reported, as it should be:", str(e)[:160])
PY
rm -f variants/libctgpu_skew.so
