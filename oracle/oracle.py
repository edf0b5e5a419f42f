"""ctypes binding of the CPU ORACLE (oracle/libctoracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (cobbletrace_amd/) never imports it.  Shares the descriptor classes of
cobbletrace_amd.bindings because the oracle takes the very same ctgpu_* structs.
"""
import ctypes as C
import os

import numpy as np

from cobbletrace_amd.bindings import FrameDesc, SceneDesc, Stats, HIT_DTYPE, RAY_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))


class Options(C.Structure):
    _fields_ = [("os_threads", C.c_int32), ("shadow_early_exit", C.c_int32), ("count_work", C.c_int32),
                ("band_first", C.c_int32), ("band_count", C.c_int32), ("band_stride", C.c_int32)]


_lib = None


def library():
    global _lib
    if _lib is None:
        path = os.environ.get("CTORACLE_LIBRARY", os.path.join(_HERE, "libctoracle.so"))     # (tools/sanitize_host.sh: the sanitizer build)
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle`")
        lib = C.CDLL(path)
        lib.cto_render.restype = C.c_int
        lib.cto_render.argtypes = [C.POINTER(SceneDesc), C.POINTER(FrameDesc), C.POINTER(Options), C.c_void_p,
                                   C.POINTER(Stats)]
        lib.cto_trace_rays.restype = C.c_int
        lib.cto_trace_rays.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_uint64, C.c_void_p]
        lib.cto_intersect_aabb.restype = C.c_int
        lib.cto_intersect_aabb.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        lib.cto_intersect_triangle.restype = C.c_int
        lib.cto_intersect_triangle.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
        lib.cto_shade_value.restype = C.c_uint32
        lib.cto_shade_value.argtypes = [C.c_uint32, C.c_float]
        lib.cto_color_to_hsv.restype = None
        lib.cto_color_to_hsv.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        lib.cto_hsv_to_color.restype = C.c_uint32
        lib.cto_hsv_to_color.argtypes = [C.POINTER(C.c_float)]
        lib.cto_blend_sample.restype = C.c_uint32
        lib.cto_blend_sample.argtypes = [C.c_uint32, C.c_uint32]
        lib.cto_average_pair.restype = C.c_uint32
        lib.cto_average_pair.argtypes = [C.c_uint32, C.c_uint32]
        lib.cto_mix_reflection.restype = C.c_uint32
        lib.cto_mix_reflection.argtypes = [C.c_uint32, C.c_uint32, C.c_float]
        lib.cto_primary_direction.restype = None
        lib.cto_primary_direction.argtypes = [C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.cto_camera_rotation.restype = None
        lib.cto_camera_rotation.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_double)]
        lib.cto_hash_rand.restype = C.c_int32
        lib.cto_hash_rand.argtypes = [C.c_uint64, C.c_uint64]
        lib.cto_fnv1a64.restype = C.c_uint64
        lib.cto_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        lib.cto_bvhtest_triangles.restype = None
        lib.cto_bvhtest_triangles.argtypes = [C.c_void_p]
        lib.cto_bvhtest_render.restype = C.c_int
        lib.cto_bvhtest_render.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


def render(scene_desc, frame, os_threads=0, shadow_early_exit=False, count_work=False, band_first=0, band_count=0,
           band_stride=1, fb=None):
    """One frame with reference semantics -> (uint32 [H, W], Stats)."""
    lib = library()
    if fb is None:
        fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
    opt = Options(os_threads, int(shadow_early_exit), int(count_work), band_first, band_count, band_stride)
    st = Stats()
    rc = lib.cto_render(C.byref(scene_desc), C.byref(frame), C.byref(opt), fb.ctypes.data, C.byref(st))
    if rc != 0:
        raise RuntimeError("cto_render failed: %d" % rc)
    return fb, st


def trace_rays(scene_desc, rays):
    lib = library()
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
    rc = lib.cto_trace_rays(C.byref(scene_desc), rays.ctypes.data, rays.shape[0], hits.ctypes.data)
    if rc != 0:
        raise RuntimeError("cto_trace_rays failed: %d" % rc)
    return hits


def fnv1a64(array):
    a = np.ascontiguousarray(array)
    return library().cto_fnv1a64(a.ctypes.data, a.nbytes)


def camera_rotation(yaw, pitch, roll):
    rot = (C.c_double * 9)()
    library().cto_camera_rotation(yaw, pitch, roll, rot)
    return rot
