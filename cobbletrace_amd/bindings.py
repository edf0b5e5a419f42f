"""ctypes bindings of libctgpu.so (include/ctgpu.h) and libcthost.so (host/capi.cpp)."""
import ctypes as C
import os

import numpy as np

# PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.  A process can
# only initialise ONE HIP runtime, and device pointers / streams handed over by torch
# (bench.py's RCCL gather) must belong to the runtime libctgpu.so talks to.  Importing torch
# first makes the dynamic loader bind libctgpu.so's NEEDED libamdhip64.so.7 to the copy
# torch already mapped (same soname) instead of mapping /opt/rocm's next to it.
try:  # plumbing only: nothing in this package computes with torch
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch-less installs use the system runtime
    torch = None

F_SUPERSAMPLE, F_SUBSAMPLE, F_SPHERES, F_REFLECT_FIX = 1, 2, 4, 8
RNG_HASH, RNG_LIBC = 0, 1
LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_AMBIENT = 0, 1, 2        # ctgpu.h CTGPU_LIGHT_* (scenefile.h:13)


class NativeLibraryMissing(RuntimeError):
    """A native library of the product is not built; nothing can be rendered."""


def repo_root():
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ descriptors (ctgpu.h)
class Material(C.Structure):
    _fields_ = [("color", C.c_uint32), ("specular", C.c_int32), ("reflection", C.c_float)]


class Node(C.Structure):
    _fields_ = [("bmin", C.c_double * 3), ("bmax", C.c_double * 3), ("left", C.c_uint32),
                ("first", C.c_uint32), ("count", C.c_uint32), ("pad_", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("intensity", C.c_float), ("position", C.c_double * 3),
                ("direction", C.c_double * 3)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_float), ("pad_", C.c_uint32), ("material", Material),
                ("pad2_", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_nodes", C.c_uint32), ("n_lights", C.c_uint32),
                ("pad_", C.c_uint32), ("vertices", C.POINTER(C.c_double)), ("materials", C.POINTER(Material)),
                ("nodes", C.POINTER(Node)), ("indexes", C.POINTER(C.c_uint32)), ("lights", C.POINTER(Light)),
                ("n_spheres", C.c_uint32), ("n_sphere_nodes", C.c_uint32), ("spheres", C.POINTER(Sphere)),
                ("sphere_nodes", C.POINTER(Node)), ("sphere_indexes", C.POINTER(C.c_uint32))]


class FrameDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("num_threads", C.c_int32), ("flags", C.c_uint32),
                ("cam_pos", C.c_double * 3), ("cam_rot", C.c_double * 9), ("rng_mode", C.c_int32),
                ("max_depth", C.c_int32), ("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("pixels_written", C.c_uint64), ("node_visits", C.c_uint64), ("triangle_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("rays_shadow_elided", C.c_uint64),
                ("walk_kind", C.c_uint32), ("bounce_kind", C.c_uint32), ("ordered", C.c_uint32), ("cull_tree", C.c_uint32),
                ("shadow_tree", C.c_uint32), ("closest_tree", C.c_uint32), ("rays_shadow_left", C.c_uint64), ("rays_primary_left", C.c_uint64)]

    def path(self):
        """which code rendered the frame, as a dict (ctgpu.h CTGPU_WALK_* / CTGPU_BOUNCE_*)"""
        return {"walk": ("general", "per-lane", "packet")[self.walk_kind], "bounce": ("none", "levels", "general")[self.bounce_kind],
                "ordered": bool(self.ordered), "cull_tree": bool(self.cull_tree), "shadow_tree": bool(self.shadow_tree),
                "closest_tree": bool(self.closest_tree), "shadow_rays_left": int(self.rays_shadow_left),
                "primary_rays_left": int(self.rays_primary_left)}

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_reflect


class WalkWork(C.Structure):
    _fields_ = [("packets", C.c_uint64 * 2), ("quad_nodes", C.c_uint64 * 2), ("triangle_records", C.c_uint64 * 2),
                ("shadow_rays_blocked", C.c_uint64), ("shadow_packets_all_blocked", C.c_uint64),
                ("rays_handed_over", C.c_uint64 * 2)]


class KernelTimes(C.Structure):
    NAMES = ("ct_sample_grid", "ct_walk_packet<primary>", "ct_light_gate", "ct_walk_packet<shadow>", "ct_shade")
    _fields_ = [("launches", C.c_uint64 * 5), ("total_ms", C.c_double * 5)]


class PhaseTimes(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("trace_ms", C.c_double), ("gather_ms", C.c_double), ("assemble_ms", C.c_double)]


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_nodes", C.c_uint32), ("tree_depth", C.c_uint32), ("n_wide", C.c_uint32),
                ("n_quads", C.c_uint32), ("n_quads_flat", C.c_uint32), ("quad_root", C.c_uint32), ("quad_root_flat", C.c_uint32),
                ("quad_pending", C.c_uint32), ("root_ref", C.c_uint32), ("slot_of_triangle0", C.c_uint32), ("shared_by", C.c_uint32),
                ("plane_bound", C.c_float * 3), ("pad_", C.c_uint32), ("build_ms", C.c_double), ("pack_ms", C.c_double),
                ("hbm_bytes", C.c_uint64), ("shadow_tree_nodes", C.c_uint32), ("shadow_tree_pending", C.c_uint32),
                ("shadow_tree_ms", C.c_double), ("shadow_tree_gamma", C.c_double)]


class Config(C.Structure):
    """ctgpu_config (include/ctgpu.h): every knob of a context; environment variable of each field in the header."""
    _fields_ = [("packet_walk", C.c_int32), ("bounce", C.c_int32), ("force_megakernel", C.c_int32), ("keep_shadow_rays", C.c_int32),
                ("blocks_per_cu", C.c_int32), ("chunk", C.c_int32), ("chunk_shadow", C.c_int32), ("lds_depth", C.c_int32),
                ("packet_lds_levels", C.c_int32), ("probe_above", C.c_int32), ("packet_budget", C.c_int32), ("packet_lanes", C.c_int32),
                ("order", C.c_int32), ("order_heavy", C.c_float * 4), ("order_floor", C.c_int32), ("order_split_below", C.c_int32),
                ("order_parts", C.c_int32 * 4), ("shadow_split_gap", C.c_float), ("big_leaf_min", C.c_int32), ("big_leaf_share", C.c_int32),
                ("shadow_tree", C.c_int32), ("closest_tree", C.c_int32), ("shadow_tree_build", C.c_int32), ("shadow_tree_leaf", C.c_int32),
                ("shadow_tree_leaf_fill", C.c_int32), ("pack_on_host", C.c_int32),
                ("thresh", C.c_int32 * 3), ("walk_thresh", C.c_int32 * 3), ("debug", C.c_uint32)]

    def as_dict(self):
        return {name: (list(getattr(self, name)) if hasattr(getattr(self, name), "__len__") else getattr(self, name)) for name, _ in self._fields_}


def default_config():
    c = Config()
    assert gpu_library().ctgpu_default_config(C.byref(c)) == 0
    return c


def config_from_environment():
    c = Config()
    assert gpu_library().ctgpu_config_from_environment(C.byref(c)) == 0
    return c


class Partition(C.Structure):
    _fields_ = [("tile_rows", C.c_int32), ("part_index", C.c_int32), ("part_count", C.c_int32)]


class Ray(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("direction", C.c_double * 3), ("t", C.c_float), ("pad_", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("tclosest", C.c_float), ("triangle", C.c_uint32), ("found", C.c_uint32)]


RAY_DTYPE = np.dtype([("origin", "<f8", 3), ("direction", "<f8", 3), ("t", "<f4"), ("pad_", "<u4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("tclosest", "<f4"), ("triangle", "<u4"), ("found", "<u4")])


WALK_GENERAL, WALK_PER_LANE, WALK_PACKET = 0, 1, 2          # ctgpu_stats.walk_kind
BOUNCE_NONE, BOUNCE_LEVELS, BOUNCE_GENERAL = 0, 1, 2        # ctgpu_stats.bounce_kind


def identity_rotation():
    """camera.rotation at yaw = pitch = roll = 0: identity with [2][0] = -0.0 (raythread.cpp:564-572)."""
    return (C.c_double * 9)(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, -0.0, 0.0, 1.0)


def camera_rotation(yaw, pitch, roll):
    """camera.rotation for the angles HandleKeyboard accumulates (raythread.cpp:564-572): products of binary32 sines and
    cosines, stored as doubles.  (numpy's float32 sin / cos may differ from glibc's in the last place; whoever compares two
    renderers hands both this one matrix.)"""
    f = np.float32
    cy, sy, cp, sp, cr, sr = (f(np.cos(f(yaw))), f(np.sin(f(yaw))), f(np.cos(f(pitch))), f(np.sin(f(pitch))), f(np.cos(f(roll))), f(np.sin(f(roll))))
    return (C.c_double * 9)(float(cy * cp), float(cy * sp * sr - sy * cr), float(cy * sp * cr + sy * sr),
                            float(sy * cp), float(sy * sp * sr + cy * cr), float(sy * sp * cr - cy * sr),
                            float(-sy), float(cp * sr), float(cp * cr))


# ------------------------------------------------------------------ library loading
_GPU_SYMBOLS = {
    # name: (restype, argtypes)
    "ctgpu_abi_version": (C.c_int, []),
    "ctgpu_build_id": (C.c_char_p, []),
    "ctgpu_last_error": (C.c_char_p, []),
    "ctgpu_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ctgpu_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ctgpu_create_with_config": (C.c_int, [C.c_int, C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "ctgpu_get_config": (C.c_int, [C.c_void_p, C.POINTER(Config)]),
    "ctgpu_default_config": (C.c_int, [C.POINTER(Config)]),
    "ctgpu_config_from_environment": (C.c_int, [C.POINTER(Config)]),
    "ctgpu_destroy": (None, [C.c_void_p]),
    "ctgpu_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "ctgpu_share_scene": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ctgpu_download_tree": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "ctgpu_scene_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "ctgpu_debug_scene_array": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ctgpu_render": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.POINTER(Stats)]),
    "ctgpu_render_begin": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p]),
    "ctgpu_render_poll": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(Stats)]),
    "ctgpu_keep_framebuffer_pinned": (C.c_int, [C.c_void_p, C.c_int]),
    "ctgpu_render_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(FrameDesc), C.c_void_p, C.POINTER(Stats)]),
    "ctgpu_partition_pixels": (C.c_int, [C.POINTER(FrameDesc), C.POINTER(Partition), C.POINTER(C.c_uint64)]),
    "ctgpu_render_partition": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.POINTER(Partition), C.c_void_p, C.c_void_p]),
    "ctgpu_assemble": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.POINTER(Partition), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ctgpu_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ctgpu_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ctgpu_comm_share": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ctgpu_comm_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ctgpu_comm_destroy": (C.c_int, [C.c_void_p]),
    "ctgpu_comm_debug_skew": (C.c_int, [C.c_void_p, C.c_int]),
    "ctgpu_gather_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]),
    "ctgpu_render_sharded": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_int, C.c_void_p, C.c_void_p]),
    "ctgpu_download_frame": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ctgpu_read_counters": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "ctgpu_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ctgpu_trace_rays_packet": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "ctgpu_build_bvh": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]),
    "ctgpu_read_walk_work": (C.c_int, [C.c_void_p, C.POINTER(WalkWork)]),
    "ctgpu_walk_choice": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ctgpu_order_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "ctgpu_bounce_choice": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ctgpu_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ctgpu_read_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(KernelTimes)]),
    "ctgpu_read_phase_times": (C.c_int, [C.c_void_p, C.POINTER(PhaseTimes)]),
    "ctgpu_count_work": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.POINTER(Stats)]),
    "ctgpu_rand_stream": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]),
}

_HOST_SYMBOLS = {
    "cth_build_id": (C.c_char_p, []),
    "cth_load_scene": (C.c_void_p, [C.c_char_p]),
    "cth_scene_from_vertices": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "cth_free_scene": (None, [C.c_void_p]),
    "cth_object_count": (C.c_int, [C.c_void_p]),
    "cth_light_count": (C.c_int, [C.c_void_p]),
    "cth_get_settings": (None, [C.c_void_p] + [C.POINTER(C.c_int)] * 4),
    "cth_set_settings": (None, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cth_get_camera": (None, [C.c_void_p, C.POINTER(C.c_double)]),
    "cth_get_rotation": (None, [C.c_void_p, C.POINTER(C.c_double)]),
    "cth_scene_desc": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "cth_tree_depth": (C.c_uint32, [C.c_void_p]),
    "cth_triangle_hash": (C.c_uint64, [C.c_void_p]),
    "cth_node_hash": (C.c_uint64, [C.c_void_p]),
    "cth_index_hash": (C.c_uint64, [C.c_void_p]),
    "cth_fnv1a64": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "cth_assemble_tiles": (C.c_int, [C.POINTER(FrameDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cth_raythread_options_async": (None, [C.c_int, C.c_uint64, C.c_int, C.c_int]),
    "cth_raythread_options_ex": (None, [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint]),
    "cth_app_loop": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "cth_blit_count": (C.c_uint64, []),
    "cth_frames_finished": (C.c_uint64, []),
    "cth_blit": (None, [C.c_void_p, C.c_int, C.c_int]),
    "cth_event_queue_script": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cth_raythread_options": (None, [C.c_int, C.c_uint64, C.c_int, C.c_int]),
    "cth_raythread_reset": (None, []),
    "cth_intersect_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "cth_move_tree_in_place": (C.c_int, [C.c_void_p, C.c_double]),
    "cth_move_one_triangle": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double]),
    "cth_raythread_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p]),
    "cth_last_frame_stats": (None, [C.POINTER(Stats)]),
}

_cache = {}


def source_build_id(which):
    """What ctgpu_build_id() / cth_build_id() of a library made from the sources ON DISK would return ("gpu" / "host"): the
    Makefiles' recipe -- sha256 over the directory's sources and headers in name order, then the public headers."""
    import glob
    import hashlib
    root = repo_root()
    if which == "gpu":
        here = os.path.join(root, "cobbletrace_amd", "csrc")
        names = sorted(os.path.basename(f) for pat in ("*.hip", "*.h") for f in glob.glob(os.path.join(here, pat)))
        public = ["ctgpu.h", "ct_tiles.h"]
    else:
        here = os.path.join(root, "cobbletrace_amd", "host")
        names = sorted(os.path.basename(f) for pat in ("*.cpp", "*.h") for f in glob.glob(os.path.join(here, pat)))
        public = ["cobbletrace_host.h", "ctgpu.h", "ct_tiles.h"]
    h = hashlib.sha256()
    for name in names:
        if name == "build_id.h":
            continue
        with open(os.path.join(here, name), "rb") as f:
            h.update(f.read())
    for name in public:
        with open(os.path.join(root, "include", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def stale_libraries():
    """[(library, built from, sources on disk)] for every native library whose build id is not that of the sources."""
    out = []
    for which, lib, fn in (("gpu", gpu_library(), "ctgpu_build_id"), ("host", host_library(), "cth_build_id")):
        built = getattr(lib, fn)().decode()
        want = source_build_id(which)
        if built != want:
            out.append((which, built, want))
    return out


def _load(path, symbols, what):
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            "%s not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()')" % (what, path))
    lib = C.CDLL(path)
    for name, (restype, argtypes) in symbols.items():
        if os.environ.get("CTGPU_OLDER_LIBRARY") and not hasattr(lib, name):
            continue                   # (tools/ab_library.py: an earlier round's library measured beside this one)
        fn = getattr(lib, name)        # AttributeError here = the library does not export what the header declares
        fn.restype = restype
        fn.argtypes = argtypes
    _cache[path] = lib
    return lib


def gpu_library_path():
    return os.environ.get("CTGPU_LIBRARY", os.path.join(repo_root(), "cobbletrace_amd", "csrc", "libctgpu.so"))


def host_library_path():
    return os.environ.get("CTHOST_LIBRARY", os.path.join(repo_root(), "cobbletrace_amd", "host", "libcthost.so"))


def gpu_library():
    return _load(gpu_library_path(), _GPU_SYMBOLS, "HIP library libctgpu.so")


def host_library():
    return _load(host_library_path(), _HOST_SYMBOLS, "host library libcthost.so")


# ------------------------------------------------------------------ host scene
class HostScene:
    """A scene parsed and BVH-built by the C++ host layer."""

    def __init__(self, path=None, vertices=None):
        self.lib = host_library()
        if path is not None:
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self.handle = C.c_void_p(self.lib.cth_load_scene(os.fsencode(path)))
        else:
            v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
            self._keep = v
            self.handle = C.c_void_p(self.lib.cth_scene_from_vertices(v.ctypes.data, v.shape[0]))
        self._desc = None

    def close(self):
        if self.handle:
            self.lib.cth_free_scene(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self):
        if self._desc is None:
            d = SceneDesc()
            self.lib.cth_scene_desc(self.handle, C.byref(d))
            self._desc = d
        return self._desc

    @property
    def settings(self):
        vals = [C.c_int() for _ in range(4)]
        self.lib.cth_get_settings(self.handle, *[C.byref(v) for v in vals])
        return dict(numberOfThreads=vals[0].value, subsampling=bool(vals[1].value), wireframe=bool(vals[2].value),
                    supersampling=bool(vals[3].value))

    def set_settings(self, threads=0, subsampling=-1, supersampling=-1):
        self.lib.cth_set_settings(self.handle, int(threads), int(subsampling), int(supersampling))

    @property
    def camera(self):
        p = (C.c_double * 3)()
        self.lib.cth_get_camera(self.handle, p)
        return tuple(p)

    def frame(self, width, height, threads=None, supersampling=None, subsampling=None, rng_mode=RNG_HASH,
              rng_seed=0, rng_offset=0, rotation=None, max_depth=10, spheres=False, reflect_fix=False):
        """FrameDesc for this scene's camera; unspecified settings come from the scene file."""
        s = self.settings
        f = FrameDesc()
        f.width, f.height = width, height
        f.num_threads = s["numberOfThreads"] if threads is None else threads
        ss = s["supersampling"] if supersampling is None else supersampling
        sub = s["subsampling"] if subsampling is None else subsampling
        f.flags = ((F_SUPERSAMPLE if ss else 0) | (F_SUBSAMPLE if sub else 0) | (F_SPHERES if spheres else 0) |
                   (F_REFLECT_FIX if reflect_fix else 0))
        f.cam_pos = (C.c_double * 3)(*self.camera)
        f.cam_rot = rotation if rotation is not None else identity_rotation()
        f.rng_mode, f.rng_seed, f.rng_offset, f.max_depth = rng_mode, rng_seed, rng_offset, max_depth
        return f

    # arrays as numpy views (owned by the handle)
    def nodes(self):
        d = self.desc
        return np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_uint8)), shape=(d.n_nodes, 64))

    def indexes(self):
        d = self.desc
        return np.ctypeslib.as_array(d.indexes, shape=(d.n_triangles,))

    def vertices(self):
        d = self.desc
        return np.ctypeslib.as_array(d.vertices, shape=(d.n_triangles, 9))


def build_bvh_on_device(vertices, device=0):
    """ctgpu_build_bvh: (nodes [m] as raw 64-byte records, indexes [n], device milliseconds) for [n][9] vertices."""
    lib = gpu_library()
    v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
    n = v.shape[0]
    nodes = np.zeros((max(2 * n - 1, 1), 64), dtype=np.uint8)
    indexes = np.zeros(n, dtype=np.uint32)
    used, ms = C.c_uint32(0), C.c_double(0.0)
    rc = lib.ctgpu_build_bvh(device, v.ctypes.data, n, nodes.ctypes.data, indexes.ctypes.data, C.byref(used), C.byref(ms))
    if rc != 0:
        raise RuntimeError("ctgpu error %d: %s" % (rc, lib.ctgpu_last_error().decode()))
    return nodes[:used.value], indexes, ms.value


# ------------------------------------------------------------------ GPU context
class GpuContext:
    """One ctgpu_ctx: a scene resident on one MI355X."""

    def __init__(self, device=0, config=None):
        """config: a Config (ctgpu_create_with_config); None: the environment's (ctgpu_create)."""
        self.lib = gpu_library()
        h = C.c_void_p()
        if config is None:
            self._check(self.lib.ctgpu_create(device, C.byref(h)))
        else:
            self._check(self.lib.ctgpu_create_with_config(device, C.byref(config), C.byref(h)))
        self.handle = h

    def get_config(self):
        c = Config()
        self._check(self.lib.ctgpu_get_config(self.handle, C.byref(c)))
        return c

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("ctgpu error %d: %s" % (rc, (self.lib.ctgpu_last_error() or b"").decode()))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ctgpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene_desc, build_on_device=False):
        """Scene -> HBM.  build_on_device: hand over vertices / materials / lights only; BuildBVH runs on the GPU."""
        if build_on_device:
            d = SceneDesc()
            C.memmove(C.byref(d), C.byref(scene_desc), C.sizeof(SceneDesc))
            d.n_nodes, d.nodes, d.indexes = 0, None, None
            scene_desc = d
        self._check(self.lib.ctgpu_upload_scene(self.handle, C.byref(scene_desc)))

    def share_scene(self, owner):
        self._check(self.lib.ctgpu_share_scene(self.handle, owner.handle))

    def scene_info(self):
        info = SceneInfo()
        self._check(self.lib.ctgpu_scene_info(self.handle, C.byref(info)))
        return info

    def download_tree(self):
        """(nodes as raw 64-byte records, indexes) of a tree that was built on the device."""
        info = self.scene_info()
        nodes = np.zeros((max(2 * info.n_triangles - 1, 1), 64), dtype=np.uint8)
        indexes = np.zeros(info.n_triangles, dtype=np.uint32)
        used = C.c_uint32(0)
        self._check(self.lib.ctgpu_download_tree(self.handle, nodes.ctypes.data, indexes.ctypes.data, C.byref(used)))
        return nodes[:used.value], indexes

    def scene_array(self, which):
        """One packed device array as bytes (test hook): 0 WideNodes, 1 QuadNodes, 2 boxes, 3 flat QuadNodes, 4 boxes, 5 TriRecords."""
        need = C.c_uint64(0)
        self._check(self.lib.ctgpu_debug_scene_array(self.handle, which, None, 0, C.byref(need)))
        out = np.zeros(need.value, dtype=np.uint8)
        self._check(self.lib.ctgpu_debug_scene_array(self.handle, which, out.ctypes.data, out.nbytes, C.byref(need)))
        return out

    def render(self, frame, fb=None):
        """Whole frame -> (uint32 [H, W] array, Stats)."""
        if fb is None:
            fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
        st = Stats()
        self._check(self.lib.ctgpu_render(self.handle, C.byref(frame), fb.ctypes.data, C.byref(st)))
        return fb, st

    @staticmethod
    def render_multi(contexts, frame, fb=None):
        """One frame split over several contexts (one per GPU) -> (uint32 [H, W], Stats)."""
        if fb is None:
            fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
        handles = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
        st = Stats()
        contexts[0]._check(contexts[0].lib.ctgpu_render_multi(handles, len(contexts), C.byref(frame), fb.ctypes.data,
                                                              C.byref(st)))
        return fb, st

    def render_begin(self, frame, fb=None):
        """Queue a whole frame into fb (allocated if omitted) and return at once; see render_poll."""
        if fb is None:
            fb = np.zeros((frame.height, frame.width), dtype=np.uint32)
        self._check(self.lib.ctgpu_render_begin(self.handle, C.byref(frame), fb.ctypes.data))
        self._in_flight_fb = fb        # the library copies into it asynchronously: it must outlive the frame
        return fb

    def render_poll(self, wait=False):
        """(done, stats) of the frame in flight."""
        done, st = C.c_int(0), Stats()
        try:
            self._check(self.lib.ctgpu_render_poll(self.handle, 1 if wait else 0, C.byref(done), C.byref(st)))
        except RuntimeError:
            self._in_flight_fb = None
            raise
        if done.value:
            self._in_flight_fb = None
        return bool(done.value), st

    def keep_framebuffer_pinned(self, keep=True):
        self._check(self.lib.ctgpu_keep_framebuffer_pinned(self.handle, 1 if keep else 0))

    def count_work(self, frame):
        st = Stats()
        self._check(self.lib.ctgpu_count_work(self.handle, C.byref(frame), C.byref(st)))
        return st

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._check(self.lib.ctgpu_trace_rays(self.handle, rays.ctypes.data, rays.shape[0], hits.ctypes.data))
        return hits

    def read_walk_work(self):
        w = WalkWork()
        self._check(self.lib.ctgpu_read_walk_work(self.handle, C.byref(w)))
        return w

    def kernel_timing(self, enable=True):
        self._check(self.lib.ctgpu_kernel_timing(self.handle, 1 if enable else 0))

    def read_kernel_times(self):
        t = KernelTimes()
        self._check(self.lib.ctgpu_read_kernel_times(self.handle, C.byref(t)))
        return t

    def walk_choice(self, supersampled):
        """(per-lane walk chosen?, last kernel ms on the shared walk, on the per-lane walk) for a frame kind."""
        lane, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
        self._check(self.lib.ctgpu_walk_choice(self.handle, 1 if supersampled else 0, C.byref(lane), C.byref(a), C.byref(b)))
        return bool(lane.value), a.value, b.value

    def bounce_choice(self, supersampled):
        """(general path chosen?, last kernel ms on the bounce pipeline, on the general path) for reflective frames of a kind."""
        general, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
        self._check(self.lib.ctgpu_bounce_choice(self.handle, 1 if supersampled else 0, C.byref(general), C.byref(a), C.byref(b)))
        return bool(general.value), a.value, b.value

    def order_info(self):
        """(list entries of the last frame's heavy classes [primary 0, primary 1, shadow 0, shadow 1], chunks cut into parts?)"""
        entries, split = (C.c_uint32 * 8)(), C.c_int(0)
        self._check(self.lib.ctgpu_order_info(self.handle, entries, C.byref(split)))
        return list(entries), bool(split.value)

    def read_phase_times(self):
        t = PhaseTimes()
        self._check(self.lib.ctgpu_read_phase_times(self.handle, C.byref(t)))
        return t

    def trace_rays_packet(self, rays, any_hit=False):
        """The per-ray hook through ct_walk_packet, the kernel the frames are traced with."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._check(self.lib.ctgpu_trace_rays_packet(self.handle, rays.ctypes.data, rays.shape[0], 1 if any_hit else 0,
                                                     hits.ctypes.data))
        return hits

    def partition_pixels(self, frame, part):
        n = C.c_uint64()
        self._check(self.lib.ctgpu_partition_pixels(C.byref(frame), C.byref(part), C.byref(n)))
        return n.value

    def render_partition(self, frame, part, d_tiles_ptr, stream_ptr=None):
        self._check(self.lib.ctgpu_render_partition(self.handle, C.byref(frame), C.byref(part), C.c_void_p(d_tiles_ptr),
                                                    C.c_void_p(stream_ptr or 0)))

    def assemble(self, frame, part, d_all_tiles_ptr, d_fb_ptr, stream_ptr=None):
        self._check(self.lib.ctgpu_assemble(self.handle, C.byref(frame), C.byref(part), C.c_void_p(d_all_tiles_ptr),
                                            C.c_void_p(d_fb_ptr), C.c_void_p(stream_ptr or 0)))

    # ---- one process per GPU: the native RCCL gather (ctgpu_comm.hip)
    @staticmethod
    def comm_unique_id():
        lib = gpu_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.ctgpu_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("ctgpu error %d: %s" % (rc, lib.ctgpu_last_error().decode()))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.ctgpu_comm_init(self.handle, buf, rank, world))

    def comm_rank(self):
        """(rank, world) as the communicator itself reports them (ncclCommUserRank / ncclCommCount where this RCCL has them)"""
        r, w = C.c_int(0), C.c_int(0)
        self._check(self.lib.ctgpu_comm_rank(self.handle, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_debug_skew(self, delta):
        self._check(self.lib.ctgpu_comm_debug_skew(self.handle, int(delta)))

    def comm_share(self, owner):
        self._check(self.lib.ctgpu_comm_share(self.handle, owner.handle))

    def gather_tiles(self, d_tiles_ptr, n, d_all_ptr, root=0, stream_ptr=None):
        self._check(self.lib.ctgpu_gather_tiles(self.handle, C.c_void_p(d_tiles_ptr), n, C.c_void_p(d_all_ptr or 0), root,
                                                C.c_void_p(stream_ptr or 0)))

    def render_sharded(self, frame, tile_rows, d_fb_ptr, stream_ptr=None):
        self._check(self.lib.ctgpu_render_sharded(self.handle, C.byref(frame), tile_rows, C.c_void_p(d_fb_ptr or 0),
                                                  C.c_void_p(stream_ptr or 0)))

    def download_frame(self, frame, d_fb_ptr, host_fb_ptr, stream_ptr=None):
        self._check(self.lib.ctgpu_download_frame(self.handle, C.byref(frame), C.c_void_p(d_fb_ptr), C.c_void_p(host_fb_ptr),
                                                  C.c_void_p(stream_ptr or 0)))

    def read_counters(self):
        st = Stats()
        self._check(self.lib.ctgpu_read_counters(self.handle, C.byref(st)))
        return st
