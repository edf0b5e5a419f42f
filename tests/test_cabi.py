"""The C-ABI library: loads, exports every symbol include/ctgpu.h declares, and behaves on a
machine without a GPU (errors, not a CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

import cobbletrace_amd as ca
from cobbletrace_amd import bindings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctgpu_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions(os.path.join(ROOT, "include", "ctgpu.h"))
    assert len(names) >= 14
    lib = C.CDLL(bindings.gpu_library_path())
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding table covers the same set
    assert sorted(bindings._GPU_SYMBOLS) == names


def test_abi_version_and_structs():
    lib = ca.gpu_library()
    assert lib.ctgpu_abi_version() == 4
    # struct sizes the header promises
    assert C.sizeof(ca.Node) == 64 and C.sizeof(ca.Light) == 56 and C.sizeof(ca.Material) == 12
    assert C.sizeof(ca.Ray) == 56 and C.sizeof(ca.Hit) == 16
    assert C.sizeof(ca.FrameDesc) == 136 and C.sizeof(ca.SceneDesc) == 88 and C.sizeof(ca.Stats) == 112


def test_no_gpu_means_errors_not_fallbacks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = ca.gpu_library()
    n = C.c_int(-1)
    rc = lib.ctgpu_device_count(C.byref(n))
    assert rc == -5 and n.value == 0
    with pytest.raises(RuntimeError, match="no HIP device|ctgpu error"):
        ca.GpuContext(0)
    assert b"HIP" in lib.ctgpu_last_error() or b"device" in lib.ctgpu_last_error()


def test_partition_geometry_is_host_side(scenes):
    """ctgpu_partition_pixels needs no device: equal tile-buffer sizes for every rank."""
    scene = scenes("cube")
    lib = ca.gpu_library()
    f = scene.frame(1920, 1080, threads=32)
    sizes = set()
    for world in (1, 2, 3, 4, 8):
        per_rank = []
        for r in range(world):
            n = C.c_uint64()
            assert lib.ctgpu_partition_pixels(C.byref(f), C.byref(ca.Partition(8, r, world)), C.byref(n)) == 0
            per_rank.append(n.value)
        assert len(set(per_rank)) == 1
        assert per_rank[0] * world >= 1056 * 1080
        sizes.add(per_rank[0])
    assert 1056 * 1080 in sizes
    n = C.c_uint64()
    assert lib.ctgpu_partition_pixels(C.byref(f), C.byref(ca.Partition(0, 0, 1)), C.byref(n)) == -1
    assert lib.ctgpu_partition_pixels(C.byref(f), C.byref(ca.Partition(8, 2, 2)), C.byref(n)) == -1
    bad = scene.frame(1920, 1080, threads=0)
    bad.num_threads = 0
    assert lib.ctgpu_partition_pixels(C.byref(bad), C.byref(ca.Partition(8, 0, 1)), C.byref(n)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("CTGPU_LIBRARY", str(tmp_path / "nope.so"))
    monkeypatch.setattr(bindings, "_cache", {})
    with pytest.raises(ca.NativeLibraryMissing):
        ca.gpu_library()


def test_product_never_touches_the_oracle():
    """The package, the native sources and the headers must not reference oracle/."""
    offenders = []
    for base in ("cobbletrace_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for name in files:
                if name.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, name), errors="ignore").read()
                    # includes, imports, library names or entry points of the oracle (prose mentions are fine)
                    if re.search(r"#include\s*[\"<][^\">]*oracle|\bimport\s+oracle|\bfrom\s+oracle|libctoracle|\bcto_[a-z]", text):
                        offenders.append(os.path.join(dirpath, name))
    assert not offenders, offenders


def test_host_library_exports_the_reference_boundary():
    """libcthost.so carries every entry point include/cobbletrace_host.h declares, with the reference's C++
    signatures (mangled names), so that the reference's own main() links against it."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "cobbletrace_host.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"^(?:bool|void|uint64_t|bvh_state_t|event_t \*)\s*\*?([A-Z][A-Za-z]+)\(", header, flags=re.M))
    assert {"RayThread", "InitSceneData", "ParseSceneFile", "ImportObject", "InitializeBVHState", "BuildBVH", "IntersectBVH",
            "IntersectBVHClosest", "ReadEvent",
            "AddEvent", "Blit", "RayThreadSetOptions", "RayThreadReset", "RayThreadBusy"} <= declared
    out = subprocess.run(["nm", "-D", "--defined-only", "--demangle", bindings.host_library_path()], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T ([A-Za-z_]+)\(", out))
    assert declared <= exported, sorted(declared - exported)
    # the signatures the reference calls with
    for signature in ("RayThread(environment_t*, scene_t*)", "ParseSceneFile(char*, scene_t*)", "Blit(environment_t*)",
                      "AddEvent(eventManager_t*, event_t)", "BuildBVH(bvh_state_t*)", "IntersectBVH(ray_t*, unsigned int, bvh_state_t*)",
                      "IntersectBVHClosest(ray_t*, unsigned int, bvh_state_t*, float*, unsigned int*)"):
        assert signature in out, signature


DEFAULT_SHADOW_SPLIT = -1.0


def test_configuration_struct_and_environment(monkeypatch):
    """ctgpu_config (include/ctgpu.h): one struct for every knob of a context.  Needs no GPU: the defaults, and what the
    environment makes of them -- each variable lands in its field, malformed values are ignored."""
    for name in list(os.environ):
        if name.startswith("CTGPU_"):
            monkeypatch.delenv(name)
    d = ca.default_config()
    assert ca.config_from_environment().as_dict() == d.as_dict()
    assert (d.packet_walk, d.bounce, d.order, d.big_leaf_min, d.big_leaf_share, d.pack_on_host, d.debug) == (-1, -1, 1, 8, 20, 0, 0)
    assert d.shadow_split_gap == DEFAULT_SHADOW_SPLIT
    assert list(d.order_heavy) == [8.0, 4.0, 2.0, 1.0] and list(d.order_parts) == [4, 2, 1, 1] and list(d.walk_thresh) == [64, 32, 24]
    env = {"CTGPU_PACKET": "0", "CTGPU_BOUNCE": "1", "CTGPU_FORCE_MEGAKERNEL": "1", "CTGPU_KEEP_SHADOW_RAYS": "1", "CTGPU_BLOCKS_PER_CU": "12",
           "CTGPU_CHUNK": "128", "CTGPU_CHUNK_SHADOW": "256", "CTGPU_LDS_DEPTH": "5", "CTGPU_PACKET_LDS_LEVELS": "9", "CTGPU_PROBE_ABOVE": "77",
           "CTGPU_PACKET_BUDGET": "96", "CTGPU_PACKET_LANES": "200", "CTGPU_ORDER": "2", "CTGPU_ORDER_HEAVY": "9,5,3,1.5", "CTGPU_ORDER_FLOOR": "3",
           "CTGPU_ORDER_SPLIT_BELOW": "4", "CTGPU_ORDER_PARTS": "8,4,2,1", "CTGPU_BIG_LEAF": "16", "CTGPU_BIG_LEAF_SHARE": "250", "CTGPU_PACK": "host",
           "CTGPU_SHADOW_SPLIT": "0.25", "CTGPU_THRESH": "1,2,3", "CTGPU_WALK_THRESH": "4,5,6", "CTGPU_DEBUG_REDO": "1", "CTGPU_DEBUG_PACKET": "1"}
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = ca.config_from_environment()
    assert (c.packet_walk, c.bounce, c.force_megakernel, c.keep_shadow_rays, c.blocks_per_cu, c.chunk, c.chunk_shadow) == (0, 1, 1, 1, 12, 128, 256)
    assert (c.lds_depth, c.packet_lds_levels, c.probe_above, c.packet_budget, c.packet_lanes) == (5, 9, 77, 96, 64)       # (lanes clamped to a wave)
    assert (c.order, c.order_floor, c.order_split_below, c.big_leaf_min, c.big_leaf_share, c.pack_on_host, c.debug) == (2, 3, 4, 16, 100, 1, 4 | 8)
    assert list(c.order_heavy) == [9.0, 5.0, 3.0, 1.5] and list(c.order_parts) == [8, 4, 2, 1]
    assert list(c.thresh) == [1, 2, 3] and list(c.walk_thresh) == [4, 5, 6] and c.shadow_split_gap == 0.25
    for k, v in {"CTGPU_ORDER_HEAVY": "1,2,3,4", "CTGPU_ORDER_PARTS": "3,3,3,3", "CTGPU_THRESH": "7", "CTGPU_PACK": "device"}.items():
        monkeypatch.setenv(k, v)                                                # rising thresholds, parts that are no power of two, ...
    c = ca.config_from_environment()
    assert list(c.order_heavy) == [8.0, 4.0, 2.0, 1.0] and list(c.order_parts) == [4, 2, 1, 1] and list(c.thresh) == [40, 20, 12] and c.pack_on_host == 0
