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
    assert lib.ctgpu_abi_version() == 3
    # struct sizes the header promises
    assert C.sizeof(ca.Node) == 64 and C.sizeof(ca.Light) == 56 and C.sizeof(ca.Material) == 12
    assert C.sizeof(ca.Ray) == 56 and C.sizeof(ca.Hit) == 16
    assert C.sizeof(ca.FrameDesc) == 136 and C.sizeof(ca.SceneDesc) == 88 and C.sizeof(ca.Stats) == 72


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
