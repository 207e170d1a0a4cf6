"""GPU tests (-m gpu) of the walks for shadow rays: 64 independent per-lane walks, one any-hit packet walk per wavefront
(packet_walk<..., ANYHIT>), or the choice between the two per wavefront.  The callers of a shadow ray read only whether it hit, so
all give the same pixels; which is fastest depends on the scene, and the library times them once per (scene, program, image)."""
import os

import numpy as np
import pytest

from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from lens_trace_amd.renderer import KERNEL_MODE_TILE, RendererHIP
from oracle import pyoracle as po
from tests.conftest import GOLDEN
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu
PATHS = {"accumulator": "examples/accumulator/resources/kernels/accumulator.cl",
         "basic_lighting": "resources/kernels/opencl/basic_lighting.cl",
         "global_illumination": "examples/global_illumination/resources/kernels/global_illumination.cl",
         "global_illumination25": "resources/kernels/opencl/global_illumination.cl"}


@pytest.fixture(scope="module")
def renderer():
    r = RendererHIP(0)
    yield r
    r.close()


def scenes():
    yield "cornell", sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    yield "lens", sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_lens_O0.ltsb")).validate() if os.path.exists(
        os.path.join(GOLDEN, "cornell_box_lens_O0.ltsb")) else synth.blob_in_box(subdiv=2).validate()
    yield "wall", synth.heightfield_wall(48).validate()
    yield "soup", synth.triangle_soup(3000).validate()
    yield "blob", synth.blob_in_box(subdiv=3).validate()


@pytest.mark.parametrize("name,scene", list(scenes()), ids=lambda v: v if isinstance(v, str) else "")
@pytest.mark.parametrize("forced", ["0", "1", "2", "3"], ids=["per-lane", "packets", "per-wavefront", "queued"])
def test_both_walks_match_the_oracle(renderer, monkeypatch, forced, name, scene):
    monkeypatch.setenv("LT_SHADOW_PACKETS", forced)
    if forced == "2":
        monkeypatch.setenv("LT_SHADOW_SPREAD", "0.05" if name in ("soup", "blob") else "0.02")   # (so that both walks occur in these small images)
    cam = sc.camera_with_frame(scene.camera, 3)
    for prog, W, H, kw in (("accumulator", 96, 64, {}), ("accumulator", 33, 17, {"kernelMode": KERNEL_MODE_TILE}),
                           ("global_illumination", 48, 40, {"giMaxDepth": 5}), ("basic_lighting", 24, 16, {}),
                           ("global_illumination25", 16, 12, {"giMaxDepth": 3})):
        for gi_path in (("1",) if not prog.startswith("global") else ("0", "1")):
            monkeypatch.setenv("LT_GI_MEGAKERNEL", gi_path)
            out = np.full((H, W, 3), np.nan, dtype=np.float32)
            renderer.render(RenderPropertiesHIP(PATHS[prog], (W, H, 3), out, scene, pCamera=cam, **kw))
            # (queued shadow rays are accumulator's: the other programs' kernels walk them per lane when told to queue)
            assert renderer.stats()["shadow_packets"] == (int(forced) if forced != "3" or prog == "accumulator" else 0)
            want = po.render(scene, cam, W, H, po.PROGRAMS[prog], kw.get("kernelMode", 0), gi_max_depth=kw.get("giMaxDepth", 16))
            assert np.array_equal(out, want), (name, prog, W, H, gi_path)


def test_the_walk_is_timed_once_per_scene_program_and_image_geometry(monkeypatch):
    monkeypatch.delenv("LT_SHADOW_PACKETS", raising=False)
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "1")
    r = RendererHIP(0)
    scene = synth.heightfield_wall(64).validate()
    W, H = 320, 200
    want = po.render(scene, sc.camera_with_frame(scene.camera, 1), W, H, po.ACCUMULATOR)

    def once(path, **kw):
        out = np.full((H, W, 3), np.nan, dtype=np.float32)
        r.render(RenderPropertiesHIP(path, (W, H, 3), out, scene, pCamera=sc.camera_with_frame(scene.camera, 1), **kw))
        return out, r.stats()

    def own(mode):                   # launches of a call that times nothing: the render launch (+ trace and resolve of queued shadow rays)
        return 3 if mode == 3 else 1

    out, st = once(PATHS["accumulator"])
    # the launch runs once untimed and once per walk (packets, per lane, chosen per wavefront, queued), then once more with the winner
    assert st["kernel_launches"] == 5 + 2 + own(st["shadow_packets"]) and st["shadow_packets"] in (0, 1, 2, 3)
    assert np.array_equal(out, want)
    chosen = st["shadow_packets"]
    out, st = once(PATHS["accumulator"])
    assert st["kernel_launches"] == own(chosen) and st["shadow_packets"] == chosen
    assert np.array_equal(out, want)
    out, st = once(PATHS["accumulator"], frameFirst=1, frameCount=3, accumulate=True)    # another number of frames per launch: timed again
    assert st["kernel_launches"] == 5 + 2 + own(st["shadow_packets"])                    # (the fold is not counted)
    fused = st["shadow_packets"]
    out, st = once(PATHS["accumulator"], frameFirst=1, frameCount=3, accumulate=True)
    assert st["kernel_launches"] == own(fused) and st["shadow_packets"] == fused          # known: nothing is repeated
    W, H = 200, 120                                                               # another image geometry: timed again, once
    want = po.render(scene, sc.camera_with_frame(scene.camera, 1), W, H, po.ACCUMULATOR)
    out, st = once(PATHS["accumulator"])
    assert st["kernel_launches"] == 5 + 2 + own(st["shadow_packets"]) and np.array_equal(out, want)
    out, st = once(PATHS["accumulator"])
    assert st["kernel_launches"] == own(st["shadow_packets"]) and np.array_equal(out, want)
    out, st = once("resources/kernels/opencl/basic.cl")
    assert st["kernel_launches"] == 1 and st["shadow_packets"] == 0               # no shadow rays: nothing to time
    out, st = once(PATHS["basic_lighting"])
    assert st["kernel_launches"] == 4 + 1                                         # its own decision (its shadow rays cannot be queued: three walks)
    r.set_scene(synth.heightfield_wall(32).validate())                            # a new scene forgets the decisions
    scene = synth.heightfield_wall(32).validate()
    out, st = once(PATHS["accumulator"], frameFirst=1, frameCount=4, accumulate=True)
    assert st["kernel_launches"] == 5 + 2 + own(st["shadow_packets"])             # timed on the fused launch itself
    out2, st = once(PATHS["accumulator"], frameFirst=1, frameCount=4, accumulate=True)
    assert st["kernel_launches"] == own(st["shadow_packets"]) and np.array_equal(out, out2)
    r.close()


def test_a_call_can_decline_the_timing_launches(monkeypatch):
    """LT_RENDER_FLAG_NO_WALK_TIMING: a first call of a geometry with the flag runs its own launch only, with the scene's most
    recent verdict for the program or any-hit packets; the pixels are the same."""
    from lens_trace_amd import _capi as C
    from lens_trace_amd.renderer import make_desc
    import torch
    monkeypatch.delenv("LT_SHADOW_PACKETS", raising=False)
    r = RendererHIP(0)
    scene = synth.heightfield_wall(48).validate()
    r.set_scene(scene)
    W, H = 160, 96
    cam = sc.camera_with_frame(scene.camera, 1)
    want = po.render(scene, cam, W, H, po.ACCUMULATOR)
    d = make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, cam, portable_math=True)
    d.flags |= C.RENDER_FLAG_NO_WALK_TIMING
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    r.render_device(d, buf.data_ptr(), buf.numel() * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = r.stats()
    assert st["kernel_launches"] == 1 and st["shadow_packets"] == 1
    assert np.array_equal(buf.cpu().numpy(), want)
    r.close()
