"""GPU tests (-m gpu) of the two scheduling features of the persistent render kernel: several samples of a running mean
in one launch (work items = (frame, square); lt_running_mean_kernel folds the per-frame images afterwards) and the
hand-out order that starts the slow-path squares (image-centre row / column) first.  Neither may change a single bit:
every case is compared with the one-launch-per-sample path (LT_FUSED_FRAMES=0, natural order) and, where a CPU oracle
run is cheap, with the oracle's accumulate (accumulator.frag:10-20)."""
import os

import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from lens_trace_amd.dist import TilePlan, untile_numpy
from lens_trace_amd.renderer import RendererHIP
from oracle import pyoracle as po
from tests.conftest import GOLDEN
from tests.conftest import oracle_desc as make_desc
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu
ACC = "examples/accumulator/resources/kernels/accumulator.cl"
GI = "examples/global_illumination/resources/kernels/global_illumination.cl"
CAM = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 1)


@pytest.fixture(scope="module")
def renderer():
    r = RendererHIP(0)
    yield r
    r.close()


@pytest.fixture(autouse=True)
def fixed_shadow_walk(monkeypatch):
    """The tests below count launches: keep the once-per-scene timing of the two shadow-ray walks (a repeated first launch)
    out of them.  test_gpu_shadow_packets.py covers it."""
    monkeypatch.setenv("LT_SHADOW_PACKETS", "0")


@pytest.fixture(scope="module")
def cornell():
    return sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()


def frames(renderer, scene, path, W, H, first, count, base=0, start=None, cam=CAM, **kw):
    out = np.full((H, W, 3), np.nan, dtype=np.float32) if start is None else start.copy()
    renderer.render(RenderPropertiesHIP(path, (W, H, 3), out, scene, pCamera=cam, frameFirst=first, frameCount=count, accumulate=True,
                                        accumulateBase=base, **kw))
    return out, renderer.stats()


@pytest.mark.parametrize("path,W,H,count", [(ACC, 96, 96, 6), (ACC, 131, 77, 5), (GI, 64, 48, 3)])
def test_fused_launch_equals_one_launch_per_sample(renderer, cornell, monkeypatch, path, W, H, count):
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "1")   # the wavefront GI pipeline has its own per-sample launches
    fused, st = frames(renderer, cornell, path, W, H, 1, count)
    assert st["kernel_launches"] == 1 and st["frames"] == count
    monkeypatch.setenv("LT_FUSED_FRAMES", "0")
    monkeypatch.setenv("LT_NATURAL_ORDER", "1")
    single, st = frames(renderer, cornell, path, W, H, 1, count)
    assert st["kernel_launches"] == count
    assert np.array_equal(fused, single)


def test_fused_launch_matches_oracle_accumulate(renderer, cornell):
    W, H = 80, 56
    got, _ = frames(renderer, cornell, ACC, W, H, 1, 7)
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for i, f in enumerate(range(1, 8)):
        po.accumulate(acc, po.render(cornell, sc.camera_with_frame(CAM, f), W, H, po.ACCUMULATOR), i)
    assert np.array_equal(got, acc)


def test_scratch_cap_splits_the_call_into_chunks(renderer, cornell, monkeypatch):
    W, H, count = 72, 40, 7
    whole, st = frames(renderer, cornell, ACC, W, H, 1, count)
    assert st["kernel_launches"] == 1
    monkeypatch.setenv("LT_FUSED_BYTES", str(3 * W * H * 3 * 4 + 100))   # room for three sample images: 3 + 3 + 1
    chunked, st = frames(renderer, cornell, ACC, W, H, 1, count)
    assert st["kernel_launches"] == 3
    assert np.array_equal(chunked, whole)
    monkeypatch.setenv("LT_FUSED_BYTES", "16")                           # not even one image: falls back to single launches
    single, st = frames(renderer, cornell, ACC, W, H, 1, count)
    assert st["kernel_launches"] == count
    assert np.array_equal(single, whole)


def test_continuing_a_running_mean_across_fused_calls(renderer, cornell):
    W, H = 64, 64
    whole, _ = frames(renderer, cornell, ACC, W, H, 1, 9)
    part, _ = frames(renderer, cornell, ACC, W, H, 1, 4)
    part, _ = frames(renderer, cornell, ACC, W, H, 5, 5, base=4, start=part)
    assert np.array_equal(part, whole)
    # frame 0 first: the reference's `if (frameCount > 0)` guard makes sample 0 an overwrite, whatever the buffer held
    a, _ = frames(renderer, cornell, ACC, W, H, 0, 3)
    b, _ = frames(renderer, cornell, ACC, W, H, 0, 3, start=np.full((H, W, 3), 123.0, dtype=np.float32))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("W,H,tile,ranks", [(200, 120, (64, 64), 3), (100, 70, (56, 16), 2), (96, 64, (96, 8), 4)])
def test_fused_tile_stacks_keep_their_padding_untouched(renderer, cornell, W, H, tile, ranks):
    import torch
    renderer.set_scene(cornell)
    whole, _ = frames(renderer, cornell, ACC, W, H, 1, 5)
    plan = TilePlan(W, H, 3, tile[0], tile[1], ranks)
    stream = torch.cuda.current_stream().cuda_stream
    stacks = []
    for r in range(ranks):
        d = make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, CAM, frame_first=1, frame_count=5, accumulate=True, accumulate_base=0,
                      tile=plan.desc_tile(r))
        buf = torch.full((plan.floats_per_rank,), -7.0, dtype=torch.float32, device="cuda:0")
        renderer.render_device(d, buf.data_ptr(), plan.floats_per_rank * 4, stream)
        torch.cuda.synchronize()
        stacks.append(buf.cpu().numpy())
    assert np.array_equal(untile_numpy(plan, stacks), whole)
    # every float of a stack is either a pixel of the image or still the sentinel
    for r in range(ranks):
        view = stacks[r].reshape(-1, plan.tile_h, plan.tile_w, 3)
        inside = np.zeros(view.shape[:3], dtype=bool)
        for k, t in enumerate(plan.tiles_of(r)):
            x0, y0, w, h = plan.tile_rect(t)
            inside[k, :h, :w] = True
        assert np.all(view[~inside] == -7.0)


@pytest.mark.parametrize("yaw", [0.0, 0.3])
def test_hand_out_order_does_not_change_pixels(renderer, monkeypatch, yaw):
    # a scene whose geometry lies in the camera's axis planes (grid lines at x = 0 and y = 2.5): the centre column / row rays
    # take the NaN-keeping box test and visit several times more nodes than their neighbours
    scene = synth.heightfield_wall(96).validate()
    cam = sc.camera_bytes(0.0, 2.5, -50.0, yaw, 0.0, 0.0, 1)
    W, H = 256, 144
    ordered, _ = frames(renderer, scene, ACC, W, H, 1, 4, cam=cam)
    monkeypatch.setenv("LT_NATURAL_ORDER", "1")
    natural, _ = frames(renderer, scene, ACC, W, H, 1, 4, cam=cam)
    assert np.array_equal(ordered, natural)
    monkeypatch.setenv("LT_PERSISTENT", "0")
    dispatched, _ = frames(renderer, scene, ACC, W, H, 1, 4, cam=cam)
    assert np.array_equal(ordered, dispatched)
    want = np.zeros((H, W, 3), dtype=np.float32)
    for i, f in enumerate(range(1, 5)):
        po.accumulate(want, po.render(scene, sc.camera_with_frame(cam, f), W, H, po.ACCUMULATOR), i)
    assert np.array_equal(ordered, want)


# ---- the wavefront GI pipeline: all frames of a call through one set of stage launches ----
@pytest.mark.parametrize("W,H,count,depth", [(64, 48, 4, 16), (97, 61, 3, 3)])
def test_wavefront_gi_fused_frames(renderer, cornell, monkeypatch, W, H, count, depth):
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "0")
    fused, st = frames(renderer, cornell, GI, W, H, 1, count, giMaxDepth=depth)
    assert st["kernel_launches"] == depth + 2          # primary + one per bounce + resolve, for all frames together
    monkeypatch.setenv("LT_FUSED_FRAMES", "0")
    single, st = frames(renderer, cornell, GI, W, H, 1, count, giMaxDepth=depth)
    assert st["kernel_launches"] == count * (depth + 2)
    assert np.array_equal(fused, single)
    want = np.zeros((H, W, 3), dtype=np.float32)
    for i, f in enumerate(range(1, count + 1)):
        po.accumulate(want, po.render(cornell, sc.camera_with_frame(CAM, f), W, H, po.GI, gi_max_depth=depth), i)
    assert np.array_equal(fused, want)


def test_wavefront_gi_fused_chunks_and_tiles(renderer, cornell, monkeypatch):
    import torch
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "0")
    W, H, count, depth = 100, 70, 5, 4
    whole, _ = frames(renderer, cornell, GI, W, H, 1, count, giMaxDepth=depth)
    per_frame = W * H * 3 * 4 + ((W + 7) // 8) * ((H + 7) // 8) * 64 * 16 * 17    # the sample image + 17 arrays of 16 bytes per path slot
    monkeypatch.setenv("LT_FUSED_BYTES", str(2 * per_frame + 64))     # two frames per chunk: 2 + 2 + 1
    chunked, st = frames(renderer, cornell, GI, W, H, 1, count, giMaxDepth=depth)
    assert st["kernel_launches"] == 3 * (depth + 2)
    assert np.array_equal(chunked, whole)
    monkeypatch.delenv("LT_FUSED_BYTES")
    plan = TilePlan(W, H, 3, 48, 32, 3)
    stream = torch.cuda.current_stream().cuda_stream
    stacks = []
    for r in range(plan.world):
        d = make_desc(C.PROGRAM_GLOBAL_ILLUMINATION, W, H, 3, CAM, frame_first=1, frame_count=count, accumulate=True, accumulate_base=0,
                      tile=plan.desc_tile(r), gi_max_depth=depth)
        buf = torch.full((plan.floats_per_rank,), -7.0, dtype=torch.float32, device="cuda:0")
        renderer.render_device(d, buf.data_ptr(), plan.floats_per_rank * 4, stream)
        torch.cuda.synchronize()
        stacks.append(buf.cpu().numpy())
    assert np.array_equal(untile_numpy(plan, stacks), whole)


# ---- the 25-sample variant: the samples of one frame through as few sets of stage launches as the scratch cap allows ----
def test_wavefront_gi25_sample_sets(renderer, cornell, monkeypatch):
    GI25 = "resources/kernels/opencl/global_illumination.cl"
    W, H, depth = 72, 40, 3
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 2)
    want = po.render(cornell, cam, W, H, po.GI25, gi_max_depth=depth)
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "0")

    def once(**kw):
        out = np.full((H, W, 3), np.nan, dtype=np.float32)
        renderer.render(RenderPropertiesHIP(GI25, (W, H, 3), out, cornell, pCamera=cam, giMaxDepth=depth, **kw))
        return out, renderer.stats()["kernel_launches"]

    got, launches = once()
    assert launches == depth + 2                       # all 25 samples in one set of launches
    assert np.array_equal(got, want)
    per_sample = W * H * 3 * 4 + ((W + 7) // 8) * ((H + 7) // 8) * 64 * 16 * 17
    monkeypatch.setenv("LT_FUSED_BYTES", str(7 * per_sample + 8))      # 7 + 7 + 7 + 4 samples per set
    got, launches = once()
    assert launches == 4 * (depth + 2)
    assert np.array_equal(got, want)
    monkeypatch.setenv("LT_FUSED_BYTES", "1")                          # one sample per set
    got, launches = once()
    assert launches == 25 * (depth + 2)
    assert np.array_equal(got, want)
    monkeypatch.delenv("LT_FUSED_BYTES")
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "1")
    got, launches = once()
    assert launches == 1 and np.array_equal(got, want)


# ---- randomised scheduling configurations: whatever the launch plan, the pixels are those of one launch per sample ----
@pytest.mark.parametrize("seed", range(int(os.environ.get("LT_FUZZ_SEEDS", "16"))))
def test_random_scheduling_configurations(renderer, cornell, monkeypatch, seed):
    import torch
    rng = np.random.default_rng(7000 + seed)
    prog_name, prog_id, path = [("accumulator", C.PROGRAM_ACCUMULATOR, ACC), ("global_illumination", C.PROGRAM_GLOBAL_ILLUMINATION, GI),
                                ("basic_lighting", C.PROGRAM_BASIC_LIGHTING, "resources/kernels/opencl/basic_lighting.cl"),
                                ("global_illumination25", C.PROGRAM_GLOBAL_ILLUMINATION_25, "resources/kernels/opencl/global_illumination.cl")][seed % 4]
    heavy = prog_id in (C.PROGRAM_BASIC_LIGHTING, C.PROGRAM_GLOBAL_ILLUMINATION_25)
    W, H = int(rng.integers(1, 40 if heavy else 150)), int(rng.integers(1, 30 if heavy else 100))
    count = int(rng.integers(2, 5 if heavy else 9))
    first = int(rng.integers(0, 20))
    base = int(rng.integers(0, 3))
    depth = int(rng.integers(1, 17))
    yaw = float(rng.choice([0.0, 0.0, 0.02]))
    cam = sc.camera_bytes(float(rng.uniform(-1, 1)) if seed % 3 else 0.0, 2.5, -50.0, yaw, 0.0, 0.0, 1)
    tile = (int(rng.integers(1, 9)) * 8, int(rng.integers(1, 9)) * 8, int(rng.integers(1, 4)))     # tile_w, tile_h, ranks
    start = rng.random((H, W, 3), dtype=np.float32)                                                # the accumulator so far (base > 0)
    renderer.set_scene(cornell)
    stream = torch.cuda.current_stream().cuda_stream

    def render_all(env):
        for k in ("LT_FUSED_FRAMES", "LT_FUSED_BYTES", "LT_NATURAL_ORDER", "LT_GI_MEGAKERNEL", "LT_PERSISTENT"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("LT_SHADOW_PACKETS", "0")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        whole = start.copy()
        renderer.render(RenderPropertiesHIP(path, (W, H, 3), whole, cornell, pCamera=cam, frameFirst=first, frameCount=count, accumulate=True,
                                            accumulateBase=base, giMaxDepth=depth))
        plan = TilePlan(W, H, 3, tile[0], tile[1], tile[2])
        stacks = []
        for r in range(plan.world):
            d = make_desc(prog_id, W, H, 3, cam, frame_first=first, frame_count=count, accumulate=True, accumulate_base=base,
                          tile=plan.desc_tile(r), gi_max_depth=depth)
            buf = torch.from_numpy(tile_stack(plan, r, start)).to("cuda:0")
            renderer.render_device(d, buf.data_ptr(), plan.floats_per_rank * 4, stream)
            torch.cuda.synchronize()
            stacks.append(buf.cpu().numpy())
        return whole, untile_numpy(plan, stacks)

    from lens_trace_amd.dist import tile_stack_numpy as tile_stack
    want, want_tiled = render_all({"LT_FUSED_FRAMES": "0", "LT_NATURAL_ORDER": "1", "LT_GI_MEGAKERNEL": "1"})
    assert np.array_equal(want, want_tiled), (prog_name, W, H, tile)
    variants = [{}, {"LT_GI_MEGAKERNEL": "0"}, {"LT_GI_MEGAKERNEL": "1"}, {"LT_SHADOW_PACKETS": "1"}, {"LT_SHADOW_PACKETS": "1", "LT_GI_MEGAKERNEL": "0"},
                {"LT_FUSED_BYTES": str(int(rng.integers(1, 6)) * (W * H * 3 * 4 + W * H * 176) + 7), "LT_GI_MEGAKERNEL": str(seed % 2)},
                {"LT_PERSISTENT": "0"}]
    for env in variants:
        got, got_tiled = render_all(env)
        assert np.array_equal(got, want), (prog_name, W, H, count, first, base, depth, env)
        assert np.array_equal(got_tiled, want), (prog_name, W, H, count, first, base, depth, tile, env)


@pytest.mark.parametrize("path", [ACC, GI])
def test_fused_launch_with_depth_4_leaves_the_fourth_channel_alone(renderer, cornell, monkeypatch, path):
    """imageDimensions[2] > 3: only channels 0..2 of a pixel are written (accumulator.cl:316-318 writes three floats at
    (y*W+x)*depth); the fold of a fused launch must neither average scratch memory into the others nor differ from one launch
    per sample."""
    W, H, count = 72, 40, 5
    monkeypatch.setenv("LT_GI_MEGAKERNEL", "1")

    def run():
        out = np.full((H, W, 4), 7.5, dtype=np.float32)
        renderer.render(RenderPropertiesHIP(path, (W, H, 4), out, cornell, pCamera=CAM, frameFirst=1, frameCount=count, accumulate=True))
        return out, renderer.stats()
    fused, st = run()
    assert st["kernel_launches"] == 1
    monkeypatch.setenv("LT_FUSED_FRAMES", "0")
    single, st = run()
    assert st["kernel_launches"] == count
    assert np.array_equal(fused, single)
    # (lt_hip_render stages through a zeroed device buffer: what was never written reads back 0, never scratch contents)
    assert np.all(fused[..., 3] == 0.0) and np.all(single[..., 3] == 0.0)
    three, _ = frames(renderer, cornell, path, W, H, 1, count)
    assert np.array_equal(fused[..., :3], three)
