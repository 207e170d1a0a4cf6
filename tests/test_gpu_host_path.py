"""GPU tests (-m gpu) of the host-buffer entry points -- the reference's contract: the caller hands over its scene with every
render() (src/opencl/renderer_opencl.cpp:107-120) and finds its HOST buffer complete on return (:146-149):

* lt_hip_render_scene (the plugin's render() with the default scene-change contract) renders the frame while the host hashes the
  scene: same pixels as lt_hip_set_scene + lt_hip_render, an unchanged scene is not uploaded again, a byte edited in place is
  honoured (the frame is rendered again);
* the read-back through the context's pinned buffer (pieces copied on by host threads) delivers the bytes a single copy does."""
import os

import numpy as np
import pytest

from lens_trace_amd import scene as sc
from lens_trace_amd.renderer import RendererHIP
from oracle import pyoracle as po
from tests.conftest import GOLDEN
from tests.conftest import oracle_props as Props

pytestmark = pytest.mark.gpu
ACC = "examples/accumulator/resources/kernels/accumulator.cl"


def test_scene_handed_over_with_every_frame_is_hashed_behind_the_frame_and_edits_are_honoured(monkeypatch):
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 2)
    W, H = 352, 288                                   # 1.2 MB of pixels: the pinned, pieced read-back
    want = po.render(s, cam, W, H, po.ACCUMULATOR)
    r = RendererHIP(0)

    def frame(**kw):
        out = np.full((H, W, 3), np.nan, dtype=np.float32)
        r.render(Props(ACC, (W, H, 3), out, s, pCamera=cam, **kw))
        return out, r.stats()

    out, st = frame()                                  # no scene yet: uploaded, then rendered
    assert np.array_equal(out, want) and st["scene_uploads"] == 1 and st["scene_reused"] == 0
    out, st = frame()                                  # the same bytes again: rendered while they are hashed, kept
    assert np.array_equal(out, want) and st["scene_uploads"] == 1 and st["scene_reused"] == 1
    mats = s.materials.view(sc.MATERIAL_DTYPE)
    old = mats["diffuse"].copy()
    mats["diffuse"][:] = old[:, ::-1]                  # an in-place edit, same sizes: the frame rendered on the old scene is dropped
    want2 = po.render(s, cam, W, H, po.ACCUMULATOR)
    assert not np.array_equal(want2, want)
    out, st = frame()
    assert np.array_equal(out, want2) and st["scene_uploads"] == 2
    out, st = frame(sceneVersion=7)                    # a versioned scene is looked at when the version is new ...
    assert np.array_equal(out, want2) and st["scene_uploads"] == 2 and st["scene_reused"] == 2
    mats["diffuse"][:] = old
    out, st = frame(sceneVersion=7)                    # ... and only then: the caller's promise
    assert np.array_equal(out, want2) and st["scene_reused"] == 2
    out, st = frame(sceneVersion=8)
    assert np.array_equal(out, want) and st["scene_uploads"] == 3
    monkeypatch.setenv("LT_PINNED_READBACK", "0")      # one copy into the caller's pageable buffer: the same bytes
    out, st = frame(sceneVersion=8)
    assert np.array_equal(out, want)
    # a running mean that continues from the caller's buffer is never rendered on an assumption
    monkeypatch.delenv("LT_PINNED_READBACK")
    acc, _ = frame(frameFirst=1, frameCount=2, accumulate=True)
    cont = acc.copy()
    r.render(Props(ACC, (W, H, 3), cont, s, pCamera=cam, frameFirst=3, frameCount=2, accumulate=True, accumulateBase=2))
    whole, _ = frame(frameFirst=1, frameCount=4, accumulate=True)
    assert np.array_equal(cont, whole)
    r.close()
