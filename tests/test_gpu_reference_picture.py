"""-m gpu twin of tests/test_reference_picture.py: the HIP frames (through the C ABI) against the two whole
frames the reference itself holds, same checks and thresholds, plus the frame-for-frame comparison with the
oracle on the very same inputs.  The seed sweep over screenshot2 (66 lights: a minute of CPU per frame) lives
here because the GPU renders it in milliseconds."""
import numpy as np
import pytest

import cobbletrace_amd as ca
from reference_picture import PICTURES, SIZE, check_frame_against_picture, outside_envelope, scene_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def picture_ctx(assets, tmp_path_factory):
    cache = {}

    def get(name):
        if name not in cache:
            scene = ca.HostScene(scene_path(name, assets, tmp_path_factory.getbasetemp()))
            ctx = ca.GpuContext(0)
            ctx.upload(scene.desc)
            cache[name] = (scene, ctx)
        return cache[name]

    yield get
    for _, ctx in cache.values():
        ctx.close()


@pytest.mark.parametrize("name", sorted(PICTURES))
def test_gpu_frame_is_the_reference_picture(name, picture_ctx):
    scene, ctx = picture_ctx(name)
    for mode, frame in (("hash", scene.frame(SIZE, SIZE, rng_mode=ca.RNG_HASH, rng_seed=0)),
                        ("libc", scene.frame(SIZE, SIZE, threads=1, rng_mode=ca.RNG_LIBC, rng_seed=1))):
        fb, st = ctx.render(frame)
        print(name, mode, check_frame_against_picture(name, fb, "gpu"), "kernel %.3f ms" % st.kernel_ms)


def test_gpu_frame_equals_the_oracle_frame_on_the_picture_scene(picture_ctx):
    """The same inputs through both: every pixel, not only the ones away from edges (<= 1 LSB; observed 0)."""
    from oracle import oracle
    scene, ctx = picture_ctx("objectLoading2")
    frame = scene.frame(SIZE, SIZE, rng_mode=ca.RNG_HASH, rng_seed=0)
    gpu_fb, _ = ctx.render(frame)
    cpu_fb, _ = oracle.render(scene.desc, frame, os_threads=8)
    shifts = np.array([0, 8, 16])
    diff = np.abs(((gpu_fb[..., None] >> shifts) & 255).astype(int) - ((cpu_fb[..., None] >> shifts) & 255).astype(int))
    print("differing pixels", int((gpu_fb != cpu_fb).sum()))
    assert diff.max() <= 1


@pytest.mark.parametrize("name,limit", [("objectLoading2", 600), ("screenshot2", 3000)])
def test_reference_picture_lies_in_the_jitter_envelope_of_gpu_frames(name, limit, picture_ctx):
    """objectLoading2: 1,955 / 852 / 331 pixels outside after 2 / 4 / 8 seeds; screenshot2: 11,726 / 5,782 /
    2,537 (oracle figures; the GPU frames are the same frames).  32 seeds take the remainder lower still."""
    scene, ctx = picture_ctx(name)
    frames, outside = [], {}
    for seed in range(32):
        frames.append(ctx.render(scene.frame(SIZE, SIZE, rng_mode=ca.RNG_HASH, rng_seed=seed))[0])
        if seed + 1 in (2, 4, 8, 32):
            outside[seed + 1] = outside_envelope(name, frames)
    print(name, outside)
    assert outside[8] <= limit
    assert outside[4] <= 0.65 * outside[2] and outside[8] <= 0.65 * outside[4] and outside[32] <= 0.5 * outside[8]
