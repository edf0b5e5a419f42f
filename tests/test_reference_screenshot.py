"""The oracle's sphere / reflection restatement against screenshot.png, as far as that picture can pin it:
tests/reference_screenshot.py says what is equal, what is not, and why (an older scene file)."""
import os

import cobbletrace_amd as ca
from oracle import oracle
from reference_screenshot import check_frame_against_screenshot, frame_for


def test_oracle_spheres_against_the_reference_screenshot(assets):
    scene = ca.HostScene(os.path.join(assets, "scene_file_1.json"))
    fb, _ = oracle.render(scene.desc, frame_for(scene), os_threads=8)
    print(check_frame_against_screenshot(fb, "oracle"))
