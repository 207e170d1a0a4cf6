"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the committed golden pixels,
the live CPU oracle and the reference's own renderer tests (tests/opencl_renderer_test.cc).

Tolerance (BASELINE.json north_star): RMS <= 1e-4 over all floats of the image.  The arithmetic is built to be
bit-identical, so each test also reports how many floats differ at all; integer counters (rays, node visits,
triangle tests) must match exactly."""
import os

import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd.renderer import KERNEL_MODE_LINEAR, KERNEL_MODE_TILE, RendererHIP
from oracle import pyoracle as po
from tests.conftest import GOLDEN, golden_index
from tests.conftest import oracle_desc as make_desc
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu
RMS_TOL = 1e-4

KERNEL_PATHS = {
    "basic": "resources/kernels/opencl/basic.cl",
    "basic_lighting": "resources/kernels/opencl/basic_lighting.cl",
    "accumulator": "examples/accumulator/resources/kernels/accumulator.cl",
    "global_illumination": "examples/global_illumination/resources/kernels/global_illumination.cl",
    "global_illumination25": "resources/kernels/opencl/global_illumination.cl",
    "custom_opencl": "examples/custom_kernel/resources/kernels/custom_opencl.cl",
}


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.fixture(scope="module")
def renderer():
    r = RendererHIP(0)
    yield r
    r.close()


_scenes = {}


def load(name):
    if name not in _scenes:
        _scenes[name] = sc.load_ltsb(os.path.join(GOLDEN, name + ".ltsb")).validate()
    return _scenes[name]


def render(renderer, scene, path, W, H, cam, mode=KERNEL_MODE_LINEAR, **kw):
    out = np.full((H, W, 3), np.nan, dtype=np.float32)
    renderer.render(RenderPropertiesHIP(path, (W, H, 3), out, scene, pCamera=cam, kernelMode=mode, **kw))
    return out


@pytest.mark.parametrize("row", golden_index(), ids=lambda r: r["tag"])
def test_golden(renderer, row):
    s = load(row["scene"])
    cam = sc.camera_bytes(0.0, 2.5, -50.0, row["yaw"], 0.0, 0.0, row["frame"])
    got = render(renderer, s, KERNEL_PATHS[row["program"]], row["W"], row["H"], cam, row["mode"], collectStats=True)
    want = np.load(os.path.join(GOLDEN, row["tag"] + ".npy"))
    ndiff = int((got != want).sum())
    print("%s: rms=%.3g floats_differing=%d/%d" % (row["tag"], rms(got, want), ndiff, want.size))
    assert not np.isnan(got).any()
    assert rms(got, want) <= RMS_TOL
    st = renderer.stats()
    for k in ("rays", "shadow_rays", "node_visits", "tri_tests"):
        assert st[k] == row[k], (k, st[k], row[k])


# ---- the reference's renderer tests, re-stated for RendererHIP (tests/opencl_renderer_test.cc) ----
CAM = sc.camera_bytes(0.0, 2.5, -50.0, 0.0)


def test_create_engine_valid_engine():
    r = RendererHIP(0)          # CreateEngineTEST.ValidEngine (:5-10)
    assert r is not None
    r.close()


def test_render_buffer_valid_buffer(renderer):
    out = render(renderer, load("green_wall_O0"), KERNEL_PATHS["basic"], 100, 100, CAM)   # :12-49
    assert not np.isnan(out).any()


def test_render_buffer_correct_color(renderer):
    out = render(renderer, load("green_wall_O0"), KERNEL_PATHS["basic"], 100, 100, CAM).reshape(-1)   # :185-228
    for x in range(0, 100 * 100, 8 * 3):
        assert out[x + 0] == 0.0 and out[x + 1] == 1.0 and out[x + 2] == 0.0
    assert np.array_equal(out.reshape(-1, 3), np.tile(np.float32([0, 1, 0]), (10000, 1)))


def test_render_buffer_kernel_mode(renderer):
    a = render(renderer, load("green_wall_O0"), KERNEL_PATHS["basic"], 100, 100, CAM, KERNEL_MODE_LINEAR).reshape(-1)
    b = render(renderer, load("green_wall_O0"), KERNEL_PATHS["basic"], 100, 100, CAM, KERNEL_MODE_TILE).reshape(-1)
    assert np.array_equal(a[::32], b[::32]) and np.array_equal(a, b)         # :120-183


def test_render_buffer_custom_block_size(renderer):
    from lens_trace_amd.renderer import THREAD_ORGANIZATION_MODE_CUSTOM, ThreadOrganizationHIP
    s = load("green_wall_O0")
    outs = [render(renderer, s, KERNEL_PATHS["basic"], 100, 100, CAM).reshape(-1)]
    for bs in [(8, 8), (4, 4)]:                                              # cuda_renderer_test.cc:88-101
        outs.append(render(renderer, s, KERNEL_PATHS["basic"], 100, 100, CAM, threadOrganizationMode=THREAD_ORGANIZATION_MODE_CUSTOM,
                           threadOrganization=ThreadOrganizationHIP(bs)).reshape(-1))
    for x in range(0, 100 * 100, 32):
        assert outs[0][x] == outs[1][x] == outs[2][x]


# ---- odd sizes, live oracle ----
@pytest.mark.parametrize("prog,W,H", [("basic", 1, 1), ("accumulator", 17, 5), ("global_illumination", 33, 47), ("basic", 130, 3)])
def test_ragged_sizes_against_live_oracle(renderer, prog, W, H):
    s = load("cornell_box_O0")
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.01, 0.0, 0.0, 5)
    got = render(renderer, s, KERNEL_PATHS[prog], W, H, cam)
    want = po.render(s, cam, W, H, po.PROGRAMS[prog])
    assert rms(got, want) <= RMS_TOL
    print(prog, W, H, "floats differing:", int((got != want).sum()))


def test_gi_depth_parameter(renderer):
    s = load("cornell_box_O0")
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 2)
    got = render(renderer, s, KERNEL_PATHS["global_illumination"], 96, 96, cam, giMaxDepth=4)
    want = po.render(s, cam, 96, 96, po.GI, gi_max_depth=4)
    assert rms(got, want) <= RMS_TOL


# ---- progressive accumulation on the device (examples/accumulator/src/main.cpp:296-325, accumulator.frag) ----
def test_running_mean_matches_oracle(renderer):
    s = load("cornell_box_O0")
    W = H = 96
    got = render(renderer, s, KERNEL_PATHS["accumulator"], W, H, CAM, frameFirst=1, frameCount=6, accumulate=True)
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for i, f in enumerate(range(1, 7)):
        po.accumulate(acc, po.render(s, sc.camera_with_frame(CAM, f), W, H, po.ACCUMULATOR), i)
    assert rms(got, acc) <= RMS_TOL
    print("running mean floats differing:", int((got != acc).sum()))
    # continuing an existing accumulator from the caller's buffer
    part = render(renderer, s, KERNEL_PATHS["accumulator"], W, H, CAM, frameFirst=1, frameCount=4, accumulate=True)
    renderer.render(RenderPropertiesHIP(KERNEL_PATHS["accumulator"], (W, H, 3), part, s, pCamera=CAM, frameFirst=5, frameCount=2,
                                        accumulate=True, accumulateBase=4))
    assert np.array_equal(part, got)


# ---- image-tile sharding + untile (the multi-GPU split, on one GPU) ----
@pytest.mark.parametrize("W,H,tile,ranks", [(200, 120, (64, 64), 3), (128, 128, (32, 16), 8), (100, 70, (100, 16), 2)])
def test_tile_sharding_reassembles_to_whole_image(renderer, W, H, tile, ranks):
    import torch
    s = load("cornell_box_O0")
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 3)
    renderer.set_scene(s)
    whole = render(renderer, s, KERNEL_PATHS["accumulator"], W, H, cam)
    descs = [make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, cam, tile=(tile[0], tile[1], r, ranks)) for r in range(ranks)]
    per_rank = max(renderer.output_floats(d) for d in descs)
    gathered = torch.zeros((ranks, per_rank), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for r, d in enumerate(descs):
        renderer.render_device(d, gathered[r].data_ptr(), per_rank * 4, stream)
    image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    renderer.untile(gathered.data_ptr(), per_rank, ranks, W, H, 3, tile[0], tile[1], image.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), whole)


# ---- error behaviour of the boundary ----
def test_errors(renderer):
    s = load("cornell_box_O0")
    out = np.zeros((10, 10, 3), dtype=np.float32)
    with pytest.raises(C.LensTraceError):
        renderer.render(RenderPropertiesHIP("resources/kernels/opencl/some_user_kernel.cl", (10, 10, 3), out, s, pCamera=CAM))
    with pytest.raises(C.LensTraceError):     # buffer too small
        renderer.render(RenderPropertiesHIP(KERNEL_PATHS["basic"], (20, 20, 3), out, s, pCamera=CAM))
    bad = sc.Scene(s.nodes.copy(), s.prims.copy(), s.materials.copy(), s.lights.copy())
    bad.node_view["offset"][0] = 10 ** 6      # right child out of range: must be refused on the host
    with pytest.raises(C.LensTraceError):
        renderer.set_scene(bad)
    fresh = RendererHIP(0)
    with pytest.raises(C.LensTraceError):     # render before set_scene
        d = make_desc(C.PROGRAM_BASIC, 4, 4, 3, CAM)
        import torch
        t = torch.zeros(48, device="cuda:0")
        fresh.render_device(d, t.data_ptr(), 48 * 4)
    fresh.close()


# ---- per-pixel integer work counters: rays, shadow rays, node visits, triangle tests, bit-exact ----
@pytest.mark.parametrize("scene,prog,W,H,frame", [
    ("cornell_box_lens_O0", "basic", 128, 128, 0),      # the image-centre ray has d.x = d.y = +0: 0*inf NaNs in the box test
    ("cornell_box_O0", "accumulator", 65, 65, 2),
    ("cornell_box_O0", "global_illumination", 64, 64, 1),
])
def test_per_pixel_counters_match_oracle_exactly(renderer, scene, prog, W, H, frame):
    s = load(scene)
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, frame)
    out = np.zeros((H, W, 4), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(KERNEL_PATHS[prog], (W, H, 4), out, s, pCamera=cam, pixelCounters=True))
    want = po.pixel_counters(s, cam, W, H, po.PROGRAMS[prog])
    assert np.array_equal(out.astype(np.uint32), want)


# ---- both execution paths of the global-illumination programs (wavefront pipeline / one lane per pixel) ----
@pytest.mark.parametrize("force", ["0", "1"], ids=["wavefront", "megakernel"])
@pytest.mark.parametrize("prog,W,H,frame,depth", [("global_illumination", 128, 128, 0, 16), ("global_illumination", 97, 61, 5, 3),
                                                  ("global_illumination", 64, 64, 2, 1), ("global_illumination25", 48, 40, 1, 16)])
def test_gi_paths_are_bit_identical_to_the_oracle(renderer, monkeypatch, force, prog, W, H, frame, depth):
    monkeypatch.setenv("LT_GI_MEGAKERNEL", force)
    s = load("cornell_box_O0")
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.01, 0.0, 0.0, frame)
    for mode in (KERNEL_MODE_LINEAR, KERNEL_MODE_TILE):
        got = render(renderer, s, KERNEL_PATHS[prog], W, H, cam, mode, giMaxDepth=depth)
        want = po.render(s, cam, W, H, po.PROGRAMS[prog], mode, gi_max_depth=depth)
        assert np.array_equal(got, want)
    # running mean through the pipeline, and tile sharding of it
    got = render(renderer, s, KERNEL_PATHS[prog], 64, 48, CAM, frameFirst=1, frameCount=3, accumulate=True, giMaxDepth=depth)
    acc = np.zeros((48, 64, 3), dtype=np.float32)
    for i, f in enumerate(range(1, 4)):
        po.accumulate(acc, po.render(s, sc.camera_with_frame(CAM, f), 64, 48, po.PROGRAMS[prog], gi_max_depth=depth), i)
    assert np.array_equal(got, acc)


@pytest.mark.parametrize("force", ["0", "1"], ids=["wavefront", "megakernel"])
def test_gi_tile_sharding(renderer, monkeypatch, force):
    import torch
    monkeypatch.setenv("LT_GI_MEGAKERNEL", force)
    s = load("cornell_box_O0")
    W, H, tile, ranks = 200, 120, (64, 64), 3
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 3)
    renderer.set_scene(s)
    whole = render(renderer, s, KERNEL_PATHS["global_illumination"], W, H, cam)
    descs = [make_desc(C.PROGRAM_GLOBAL_ILLUMINATION, W, H, 3, cam, tile=(tile[0], tile[1], r, ranks)) for r in range(ranks)]
    per_rank = max(renderer.output_floats(d) for d in descs)
    gathered = torch.zeros((ranks, per_rank), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for r, d in enumerate(descs):
        renderer.render_device(d, gathered[r].data_ptr(), per_rank * 4, stream)
    image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    renderer.untile(gathered.data_ptr(), per_rank, ranks, W, H, 3, tile[0], tile[1], image.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), whole)


def test_in_place_scene_edits_show_without_any_invalidate_call(renderer):
    """The reference uploads every scene buffer on every render() (src/opencl/renderer_opencl.cpp:107-120).  Here the resident
    copy is kept only while lt_hip_set_scene's hash of EVERY byte is unchanged: an edit of any single vertex / material shows."""
    import copy
    s = copy.deepcopy(load("cornell_box_O0"))
    before = renderer.stats()
    a = render(renderer, s, KERNEL_PATHS["basic"], 64, 64, CAM)
    b = render(renderer, s, KERNEL_PATHS["basic"], 64, 64, CAM)
    st = renderer.stats()
    assert np.array_equal(a, b)
    # (the copy has the content of a scene an earlier test may have left resident: at most one upload, at least one reuse)
    assert st["scene_uploads"] <= before["scene_uploads"] + 1 and st["scene_reused"] >= before["scene_reused"] + 1
    # one float of one material, somewhere in the middle of the buffer (a strided sample would miss it)
    mats = s.materials.view(np.float32).reshape(-1, 8)
    mats[len(mats) // 2, 1] += 0.25
    c = render(renderer, s, KERNEL_PATHS["basic"], 64, 64, CAM)
    assert renderer.stats()["scene_uploads"] == st["scene_uploads"] + 1
    assert not np.array_equal(a, c)
    assert np.array_equal(c, po.render(s, CAM, 64, 64, po.BASIC))
    # a versioned scene is not looked at again until its version changes
    d = render(renderer, s, KERNEL_PATHS["basic"], 64, 64, CAM, sceneVersion=3)
    mats[len(mats) // 2, 1] -= 0.25
    e = render(renderer, s, KERNEL_PATHS["basic"], 64, 64, CAM, sceneVersion=3)
    f = render(renderer, s, KERNEL_PATHS["basic"], 64, 64, CAM, sceneVersion=4)
    assert np.array_equal(d, c) and np.array_equal(e, c) and np.array_equal(f, a)
