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


def test_edits_that_leave_the_nodes_alone_keep_the_hierarchies(monkeypatch):
    """A material, a light list or the primitives edited in place (same sizes, nodes untouched): the changed buffers are uploaded
    and the leaf records re-made on the device -- no host-side build (own_tree_ms keeps the value of the one build) -- and the
    frame equals the oracle's on the edited scene, through the packet walks, the per-lane walks and the queued shadow rays."""
    from lens_trace_amd import synth
    s = synth.heightfield_wall(40).validate()
    cam = sc.camera_with_frame(s.camera, 3)
    W, H = 160, 96
    r = RendererHIP(0)

    def frame():
        out = np.full((H, W, 3), np.nan, dtype=np.float32)
        r.render(Props(ACC, (W, H, 3), out, s, pCamera=cam))
        return out, r.stats()

    for mode in ("1", "0", "3"):
        monkeypatch.setenv("LT_SHADOW_PACKETS", mode)
        out, st = frame()
        built = st["own_tree_ms"]
        uploads = st["scene_uploads"]
        assert np.array_equal(out, po.render(s, cam, W, H, po.ACCUMULATOR)) and st["own_tree_height"] > 0
        prims = s.prims.view(sc.PRIM_DTYPE)
        keep = prims.copy()
        # primitives: flip the normals of a third of the wall and move a vertex of every 7th triangle (inside its leaf's box or
        # not: the box is the caller's business, the triangle test is the reference's on the new vertices)
        prims["normalA"][::3] *= -1.0
        prims["normalB"][::3] *= -1.0
        prims["normalC"][::3] *= -1.0
        prims["positionB"][::7, 2] -= 0.01
        out, st = frame()
        assert st["scene_uploads"] == uploads + 1 and st["own_tree_ms"] == built
        assert np.array_equal(out, po.render(s, cam, W, H, po.ACCUMULATOR))
        mats = s.materials.view(sc.MATERIAL_DTYPE)
        mats["diffuse"][:] = mats["diffuse"][:, ::-1]
        out, st = frame()
        assert st["scene_uploads"] == uploads + 2 and st["own_tree_ms"] == built
        assert np.array_equal(out, po.render(s, cam, W, H, po.ACCUMULATOR))
        prims[:] = keep
        out, st = frame()
        assert st["scene_uploads"] == uploads + 3 and np.array_equal(out, po.render(s, cam, W, H, po.ACCUMULATOR))
        # a node edited in place is another matter: everything is built again
        nodes = s.nodes.view(sc.NODE_DTYPE)
        leaf = int(np.flatnonzero(nodes["primitiveCount"] != 0)[5])
        saved = nodes["boundsMax"][leaf].copy()
        nodes["boundsMax"][leaf] = nodes["boundsMin"][leaf]            # the leaf's box shrinks to a point: its triangle is missed as the reference misses it
        out, st = frame()
        assert st["scene_uploads"] == uploads + 4 and np.array_equal(out, po.render(s, cam, W, H, po.ACCUMULATOR))
        nodes["boundsMax"][leaf] = saved
        out, st = frame()
        assert np.array_equal(out, po.render(s, cam, W, H, po.ACCUMULATOR))
    r.close()


def test_a_scene_that_changes_with_every_frame_is_hashed_before_the_frame():
    """An animation: after a call that found the scene changed, the next call looks at the scene first (no frame rendered on the
    old one for nothing); after a call that found it unchanged, frames start at once again.  Either way the pixels are the
    scene's that came with the frame."""
    a = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    b = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    mb = b.materials.view(sc.MATERIAL_DTYPE)
    mb["diffuse"][:] = mb["diffuse"][:, ::-1].copy()
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 2)
    W, H = 160, 120
    want = {id(a): po.render(a, cam, W, H, po.ACCUMULATOR), id(b): po.render(b, cam, W, H, po.ACCUMULATOR)}
    assert not np.array_equal(want[id(a)], want[id(b)])
    r = RendererHIP(0)
    uploads = 0
    last = None
    for s in (a, b, a, b, b, b, a, a, b):
        out = np.full((H, W, 3), np.nan, dtype=np.float32)
        r.render(Props(ACC, (W, H, 3), out, s, pCamera=cam))
        uploads += 0 if s is last else 1
        last = s
        st = r.stats()
        assert np.array_equal(out, want[id(s)]) and st["scene_uploads"] == uploads
    r.close()
