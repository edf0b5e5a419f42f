"""The shadow tree's builder on the CPU (cobbletrace_amd/csrc/ct_shadow_tree_build.h is plain C++): tests/native/
shadow_tree_check.cpp is compiled against it with g++ and run over triangle-box layouts that try the builder's branches --
soups, flat sheets, coincident boxes (no plane separates them: median splits), exponentially spaced boxes (the depth
bound), big overlapping boxes -- for both split strategies, one to four triangles a leaf, one thread and several.  The
harness checks what the walk relies on: a tree (no node reached twice), every triangle under exactly one entry, every
entry's box around what lies below it by gamma, entries a node does not have at the far corner with a reference that is
safe to follow, the stack bound the builder reports, and -- walking it with the kernel's own stack discipline -- that a ray
reaches every triangle whose box its line enters.  (What the tree is for: DESIGN appendix E; the kernel's side: -m gpu,
tests/test_gpu_shadow_tree.py.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("stree") / "shadow_tree_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "cobbletrace_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "shadow_tree_check.cpp"), "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("layout", [0, 1, 2, 3, 4])
def test_builder_contract(harness, layout):
    for n in (1, 2, 3, 4, 5, 9, 17, 300, 5000):
        for strategy in (0, 1, 2):             # 2: a tree as the device's clustering makes them (restated in the harness), through the host's four-wide pass
            for leaf, fill in ((1, 0), (2, 50), (4, 70)):
                out = subprocess.run([harness, str(n), str(11 + n), str(strategy), str(layout), "1", str(leaf), str(fill)],
                                     capture_output=True, text=True, timeout=120)
                assert out.returncode == 0 and out.stdout.startswith("OK"), (n, strategy, layout, leaf, fill, out.stdout, out.stderr[-2000:])


def test_builder_on_several_threads(harness):
    for layout, leaf, strategy in ((0, 2, 0), (4, 4, 0), (0, 2, 2)):
        out = subprocess.run([harness, "140000", "5", str(strategy), str(layout), "6", str(leaf), "30"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.startswith("OK"), (out.stdout, out.stderr[-2000:])
