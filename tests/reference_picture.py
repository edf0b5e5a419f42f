"""The two whole frames the REFERENCE itself holds: objectLoading2.png and screenshot2.png.

Both are byte copies of files in the reference tree that its README shows (README.md:23-24) -- the
author's screenshots of the reference at work in its own 640x640 window (`cobbletrace.cpp:14-15`).
They are data (pixels), not source:

  tests/golden/reference_objectLoading2.png   sha256 f55d657b...547b   642x643 (window frame of 1-2 px);
      rows 0..639, columns 2..641 are the frame of `scene_import.json` with the scene file's settings
      (32 workers, supersampling on): four OBJ meshes, 1,712 triangles, three lights.
  tests/golden/reference_screenshot2.png      sha256 14985757...4c7d   640x637; its rows are rows 1..637
      of the frame of `utils/pc_big.json` (kept as tests/golden/pc_big.json.gz, a scene DATA file: 6,468
      inline triangles, 80 spheres the reference does not draw, 66 lights; 16 workers, supersampling on).
      Found by rendering the scene files under utils/ and sliding the picture over the frame.

What a frame of ours can and cannot share with them.  Everything except the supersampling jitter is
deterministic: parser, loader, placement, BVH, traversal, lighting over every light, pow, HSV, the
16-sample running blend.  The jitter is `rand()` drawn by racing workers (`raythread.cpp:470-471`), a
stream nobody can replay (MSVC's per-thread generator was tried too: worse), so pixels on silhouette edges
and on the steps of a colour ramp may differ; everywhere else they must not.  Checks per picture
(thresholds from the oracle's measured figures, listed in PICTURES):
  * the written-pixel mask is the reference's (row 0 is never traced);
  * a floor on the bit-equal fraction and on the fraction within 2 LSB per channel;
  * every pixel whose neighbourhood in the picture is ONE colour (no edge, no ramp step) is exact;
  * every pixel whose neighbourhood spans at most 1 LSB per channel (a ramp, no edge) is within 1 LSB;
  * the picture lies inside the envelope of our own frames over several jitter seeds (per pixel and
    channel, min..max) except for a fraction that shrinks as seeds are added, as jitter noise must.
"""
import gzip
import os
import shutil

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZE = 640
_SHIFTS = np.array([0, 8, 16])

PICTURES = {
    # picture rows/cols that hold the frame, and which frame rows those are.
    # measured with the oracle (hash jitter, seed 0): 99.05 % equal, 99.70 % within 2
    "objectLoading2": dict(png="reference_objectLoading2.png", shape=(643, 642, 4), pic_rows=(0, 640),
                           pic_cols=(2, 642), frame_rows=(0, 640), scene="scene_import.json", neighbourhood=5,
                           min_equal=0.985, min_close=0.995, flat_pixels=262395, flat_on_objects=34583),
    # 94.91 % equal, 98.56 % within 2 (a lattice of 6,468 small faces: edges everywhere)
    "screenshot2": dict(png="reference_screenshot2.png", shape=(637, 640, 4), pic_rows=(0, 637),
                        pic_cols=(0, 640), frame_rows=(1, 638), scene="pc_big.json", neighbourhood=3,
                        min_equal=0.94, min_close=0.98, flat_pixels=234181, flat_on_objects=15116),
}
BACKGROUND = 0x333333


def scene_path(name, assets, tmp_dir):
    """Path of the picture's scene file (pc_big.json is unpacked into tmp_dir)."""
    scene = PICTURES[name]["scene"]
    if scene == "pc_big.json":
        out = os.path.join(str(tmp_dir), "pc_big.json")
        if not os.path.exists(out):
            with gzip.open(os.path.join(GOLDEN, "pc_big.json.gz"), "rb") as src, open(out, "wb") as dst:
                shutil.copyfileobj(src, dst)
        return out
    return os.path.join(assets, scene)


def load_reference_frame(name):
    """(the picture's part of the frame as the reference's framebuffer words 0x00BBGGRR -- draw2d.h:8-20,
    the slice of frame rows it shows)."""
    from PIL import Image
    pic = PICTURES[name]
    rgba = np.array(Image.open(os.path.join(GOLDEN, pic["png"])))
    assert rgba.shape == pic["shape"]
    crop = rgba[slice(*pic["pic_rows"]), slice(*pic["pic_cols"]), :3].astype(np.uint32)
    return crop[..., 0] | (crop[..., 1] << 8) | (crop[..., 2] << 16), slice(*pic["frame_rows"])


def channels(fb):
    return ((fb[..., None] >> _SHIFTS) & 255).astype(np.int32)


def neighbourhood_span(fb, size):
    """Per pixel: the largest per-channel (max - min) over its size x size neighbourhood."""
    from scipy.ndimage import maximum_filter, minimum_filter
    ch = channels(fb)
    span = np.zeros(fb.shape, dtype=np.int32)
    for c in range(3):
        span = np.maximum(span, maximum_filter(ch[..., c], size=size) - minimum_filter(ch[..., c], size=size))
    return span


def check_frame_against_picture(name, fb, what):
    """All the single-frame checks of the module docstring; returns the figures for the log."""
    pic = PICTURES[name]
    png, rows = load_reference_frame(name)
    assert fb.shape == (SIZE, SIZE)
    assert (fb[0] == 0).all(), "%s: row 0 is never traced (raythread.cpp:182)" % what
    assert int((fb != 0).sum()) == SIZE * (SIZE - 1)
    ours = fb[rows]
    written = ours != 0
    assert ((png != 0) == written).all(), "%s: written-pixel mask differs from the reference's" % what
    diff = np.abs(channels(ours) - channels(png)).max(-1)
    equal = float((diff[written] == 0).mean())
    close = float((diff[written] <= 2).mean())
    assert equal >= pic["min_equal"], "%s: only %.4f of the written pixels are bit-equal" % (what, equal)
    assert close >= pic["min_close"], "%s: only %.4f of the written pixels are within 2 LSB" % (what, close)
    span = neighbourhood_span(png, pic["neighbourhood"])
    flat = (span == 0) & written
    ramp = (span <= 1) & written
    assert int(flat.sum()) == pic["flat_pixels"]
    assert int((flat & (png != BACKGROUND)).sum()) == pic["flat_on_objects"]
    assert int((diff[flat] != 0).sum()) == 0, "%s: %d pixels away from any edge differ" % (what, (diff[flat] != 0).sum())
    assert int(diff[ramp].max()) <= 1, "%s: a ramp pixel is off by %d" % (what, diff[ramp].max())
    return {"bit_equal": equal, "within_2": close, "flat_pixels": int(flat.sum()), "ramp_pixels": int(ramp.sum()),
            "ramp_exact": float((diff[ramp] == 0).mean())}


def outside_envelope(name, frames):
    """Pixels of the picture that lie outside [min, max] of `frames` in some channel."""
    png, rows = load_reference_frame(name)
    png = channels(png)
    stack = np.stack([channels(f[rows]) for f in frames])
    lo, hi = stack.min(0), stack.max(0)
    return int(((png < lo) | (png > hi)).any(-1).sum())
