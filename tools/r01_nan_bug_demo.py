"""One-off evidence for VERDICT r01 weak #1: render the `tables` scene of
tests/test_gpu_parity.py::test_nan_parameter_rays_that_hit with the library given in CTGPU_LIBRARY
(e.g. a build of the round-1 source) and count the pixels that differ from the oracle."""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cobbletrace_amd as ca                      # noqa: E402
from cobbletrace_amd import bindings              # noqa: E402

if os.environ.get("CTGPU_LIBRARY"):
    bindings._GPU_SYMBOLS.pop("ctgpu_trace_rays_packet", None)     # the round-1 library has no such symbol
import test_gpu_parity as T                       # noqa: E402
from oracle import oracle                         # noqa: E402

scene = ca.HostScene(T._tables_scene(pathlib.Path(tempfile.mkdtemp())))
ctx = ca.GpuContext(0)
ctx.upload(scene.desc)
for ss in (False, True):
    frame = scene.frame(240, 160, threads=4, supersampling=ss, rng_seed=2)
    got, _ = ctx.render(frame)
    want, _ = oracle.render(scene.desc, frame, os_threads=2)
    print("library %s ss=%d: %d of %d pixels differ from the oracle (centre row: %d)" %
          (os.environ.get("CTGPU_LIBRARY", "HEAD"), ss, int((got != want).sum()), got.size,
           int((got[80] != want[80]).sum())))
