"""-m gpu: the HIP frame with CTGPU_F_SPHERES | CTGPU_F_REFLECT_FIX against screenshot.png (what that picture can pin:
tests/reference_screenshot.py) and, pixel for pixel, against the oracle's frame of the same flags."""
import os

import pytest

import cobbletrace_amd as ca
from oracle import oracle
from reference_screenshot import check_frame_against_screenshot, frame_for

pytestmark = pytest.mark.gpu


def test_hip_spheres_against_the_reference_screenshot(assets):
    scene = ca.HostScene(os.path.join(assets, "scene_file_1.json"))
    ctx = ca.GpuContext(0)
    try:
        ctx.upload(scene.desc)
        frame = frame_for(scene)
        fb, _ = ctx.render(frame)
        print(check_frame_against_screenshot(fb, "HIP"))
        want, _ = oracle.render(scene.desc, frame, os_threads=8)
        assert (fb == want).all(), int((fb != want).sum())
    finally:
        ctx.close()
