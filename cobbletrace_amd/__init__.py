"""cobbletrace_amd -- MI355X-native ray-trace hot path of CobbleTrace.

The product is two native libraries:

* ``csrc/libctgpu.so``  -- hand-written gfx950 kernels behind the C ABI of ``include/ctgpu.h``
* ``host/libcthost.so`` -- C++ host layer with the reference's input API
  (``InitSceneData`` / ``ParseSceneFile`` / ``ImportObject`` / ``BuildBVH``) and the
  ``RayThread(env, scene)`` drop-in

This Python package is plumbing for the test and bench harness only: ctypes bindings of
those two libraries.  It never computes pixels itself and it raises if the HIP library
is missing -- there is no CPU fallback.
"""
from .bindings import (  # noqa: F401
    FrameDesc, Hit, Light, Material, Node, Partition, Ray, SceneDesc, Sphere, Stats, WalkWork, KernelTimes, PhaseTimes, SceneInfo,
    GpuContext, HostScene, NativeLibraryMissing, Config, default_config, config_from_environment, camera_rotation,
    F_SUPERSAMPLE, F_SUBSAMPLE, F_SPHERES, F_REFLECT_FIX, RNG_HASH, RNG_LIBC, LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_AMBIENT,
    build_bvh_on_device, gpu_library, host_library, identity_rotation, repo_root,
)
