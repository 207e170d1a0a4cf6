"""GPU tests (-m gpu) at BASELINE.json's full sizes: the 1 002 530-triangle synthetic scene at 3840x2160.
The CPU oracle needs minutes per 4K frame, so whole frames are checked through size-independent properties, and the
oracle itself is consulted on sampled rows:
  * rows sampled across the frame == oracle rows, bit for bit (same buffers, camera, frameCount);
  * tile-sharded render (8 ranks' worth, rendered one after another on this GPU) + untile == whole-frame render;
  * the 16-sample device-side running mean == folding 16 single-frame renders with accumulator.frag's formula;
  * tileKernel output clamped == linearKernel output;
  * ray bookkeeping: rays == pixels + shadow rays, every pixel rendered once;
  * the WHOLE 4K frame, in the default flavour (what bench.py times), == the reference's own accumulator.cl compiled for
    gfx950 (oracle/ref_gpu.py) run on the same 1 M-triangle buffers, bit for bit, with either shadow-ray walk forced, camera
    unrotated and rotated."""
import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from lens_trace_amd.dist import TilePlan
from lens_trace_amd.renderer import KERNEL_MODE_TILE, RendererHIP
from oracle import pyoracle as po
from tests.conftest import oracle_desc as make_desc
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu
W, H = 3840, 2160
ACC = "examples/accumulator/resources/kernels/accumulator.cl"


@pytest.fixture(scope="module")
def wall():
    return synth.heightfield_wall(708).validate()


@pytest.fixture(scope="module")
def renderer(wall):
    r = RendererHIP(0)
    r.set_scene(wall)
    yield r
    r.close()


def render(renderer, scene, frame, **kw):
    out = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(ACC, (W, H, 3), out, scene, pCamera=sc.camera_with_frame(scene.camera, frame), **kw))
    return out


def test_scene_is_the_1m_triangle_config(wall):
    assert wall.n_prims == 1002530 and wall.n_nodes == 2 * 1002530 - 1
    assert wall.light_view["count"][0] == 2
    assert wall.node_view["primitiveCount"].max() == 1     # own builder: one triangle per leaf


def test_sampled_rows_match_oracle_bit_for_bit(renderer, wall):
    got = render(renderer, wall, 3)
    cam = sc.camera_with_frame(wall.camera, 3)
    for y in (0, 1, 537, 1079, 1080, 1081, 1620, 2159):
        want = po.render(wall, cam, W, H, po.ACCUMULATOR, rows=(y, y + 1), threads=1)
        assert np.array_equal(got[y], want[y]), "row %d" % y


def test_work_counters_and_ray_bookkeeping(renderer, wall):
    out = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(ACC, (W, H, 3), out, wall, pCamera=sc.camera_with_frame(wall.camera, 1), collectStats=True))
    st = renderer.stats()
    assert st["pixels"] == W * H
    assert st["rays"] == W * H + st["shadow_rays"]
    assert st["node_visits"] > 100 * st["rays"] and st["tri_tests"] >= st["rays"] // 2
    # the oracle's counters on a band of rows equal the device's on the same band (tile = 3840 x 8 rows, tile 135)
    d = make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, sc.camera_with_frame(wall.camera, 1), tile=(W, 8, 135, 10 ** 6), stats=True)
    import torch
    buf = torch.zeros(renderer.output_floats(d), dtype=torch.float32, device="cuda:0")
    renderer.render_device(d, buf.data_ptr(), buf.numel() * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    band = renderer.stats()
    _, ost = po.render(wall, sc.camera_with_frame(wall.camera, 1), W, H, po.ACCUMULATOR, rows=(1080, 1088), threads=4, want_stats=True)
    for k in ("rays", "shadow_rays", "node_visits", "tri_tests"):
        assert band[k] == ost[k], k
    assert np.array_equal(buf.cpu().numpy().reshape(8, W, 3), out[1080:1088])


def test_tile_sharding_of_the_4k_frame_reassembles_exactly(renderer, wall):
    import torch
    whole = render(renderer, wall, 2)
    plan = TilePlan(W, H, 3, 64, 64, 8)
    stream = torch.cuda.current_stream().cuda_stream
    stack = torch.zeros((8, plan.floats_per_rank), dtype=torch.float32, device="cuda:0")
    for r in range(8):
        d = make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, sc.camera_with_frame(wall.camera, 2), tile=plan.desc_tile(r))
        assert renderer.output_floats(d) == plan.floats_per_rank
        renderer.render_device(d, stack[r].data_ptr(), plan.floats_per_rank * 4, stream)
    image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    renderer.untile(stack.data_ptr(), plan.floats_per_rank, 8, W, H, 3, 64, 64, image.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), whole)


def test_bench_partition_of_the_16_sample_frame_reassembles_exactly(renderer, wall):
    """What `bench.py --gpus 8` computes: every rank's balanced-plan share of the 16-sample running mean in one fused launch,
    gathered and untiled, equals the whole frame rendered on one GPU."""
    import torch
    whole = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(ACC, (W, H, 3), whole, wall, pCamera=wall.camera, frameFirst=1, frameCount=16, accumulate=True))
    plan = TilePlan.balanced(W, H, 3, 8)
    assert plan.tiles_x % 2 == 1          # coprime to 8 ranks: every tile column is spread over all of them
    stream = torch.cuda.current_stream().cuda_stream
    stack = torch.zeros((8, plan.floats_per_rank), dtype=torch.float32, device="cuda:0")
    for r in range(8):
        d = make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, wall.camera, frame_first=1, frame_count=16, accumulate=True, accumulate_base=0,
                      tile=plan.desc_tile(r))
        renderer.render_device(d, stack[r].data_ptr(), plan.floats_per_rank * 4, stream)
        assert renderer.stats()["kernel_launches"] >= 1           # (more than the call's own when it is the first of its geometry: the shadow-ray walks are timed)
    image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    renderer.untile(stack.data_ptr(), plan.floats_per_rank, 8, W, H, 3, plan.tile_w, plan.tile_h, image.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), whole)


def test_shadow_ray_walks_agree_on_the_full_frame(renderer, wall, monkeypatch):
    """The any-hit packet walk, the per-lane walk and the per-wavefront choice between them for shadow rays (the library picks
    one per scene by timing) give the same 4K frame, with a rotated camera too (mixed direction signs inside wavefronts)."""
    for yaw in (0.0, 0.4):
        cam = sc.camera_bytes(0.0, 2.5, -50.0, yaw, 0.0, 0.0, 3)
        frames = {}
        monkeypatch.setenv("LT_SHADOW_SPREAD", "0.004")     # (the wall's bundles sit around this spread: both walks occur)
        for mode in ("0", "1", "2", "3"):
            monkeypatch.setenv("LT_SHADOW_PACKETS", mode)
            out = np.empty((H, W, 3), dtype=np.float32)
            renderer.render(RenderPropertiesHIP(ACC, (W, H, 3), out, wall, pCamera=cam, frameFirst=3, frameCount=2, accumulate=True))
            assert renderer.stats()["shadow_packets"] == int(mode)
            frames[mode] = out
        assert np.array_equal(frames["0"], frames["1"]) and np.array_equal(frames["0"], frames["2"])


def test_running_mean_of_16_frames_equals_folding_single_frames(renderer, wall):
    got = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(ACC, (W, H, 3), got, wall, pCamera=wall.camera, frameFirst=1, frameCount=16, accumulate=True))
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for i, f in enumerate(range(1, 17)):
        po.accumulate(acc.reshape(-1), render(renderer, wall, f).reshape(-1), i)
    assert np.array_equal(got, acc)
    # and it is a mean: within float rounding of the plain average, pixel values in [0, 1]
    assert got.min() >= 0.0 and got.max() <= 1.0


def test_tile_mode_clamped_equals_linear_mode(renderer, wall):
    lin = render(renderer, wall, 5)
    til = render(renderer, wall, 5, kernelMode=KERNEL_MODE_TILE)
    assert np.array_equal(np.clip(til, 0.0, 1.0), lin)
    assert til.min() < 0.0      # back-facing n.l survives in tile mode


# ---- the timed kernel against the reference itself, at full size --------------------------------------------------------
@pytest.mark.parametrize("yaw,build", [(0.0, "default"), (0.03, "default"), (0.0, "strict")])
def test_whole_4k_frame_matches_reference_accumulator_kernel(renderer, wall, monkeypatch, yaw, build):
    """examples/accumulator/resources/kernels/accumulator.cl:113-217 (one work-item per pixel, private 64-entry stack) on the
    1 002 530-triangle buffers at 3840x2160, built with the reference's own (NULL) build options, against the HIP path's default
    flavour -- the kernel bench.py times: the backend's own hierarchy, hand-written packet walks over its pair records (stack in
    one register), octant switches, slow-path squares first, and the shadow rays through each of their walks -- as packets, per
    lane over the 4-wide groups, queued for lt_trace_kernel.  Bit for bit.  (And the strict flavour against the strict build of the same file.)"""
    from lens_trace_amd.renderer import RenderPropertiesHIP as DefaultFlavourProps
    from oracle import ref_gpu
    if not ref_gpu.available("accumulator", build):
        pytest.skip("oracle/_ref/accumulator.%s.co not built (needs /root/reference at build time)" % build)
    cam = sc.camera_bytes(0.0, 2.5, -50.0, yaw, 0.0, 0.0, 2)
    ref = ref_gpu.render(wall, cam, W, H, "accumulator", build)
    assert ref.shape == (H, W, 3) and ref.sum() > 0
    for packets in ("1", "0", "3"):
        monkeypatch.setenv("LT_SHADOW_PACKETS", packets)
        got = np.empty((H, W, 3), dtype=np.float32)
        renderer.render(DefaultFlavourProps(ACC, (W, H, 3), got, wall, pCamera=cam, strictMath=(build == "strict")))
        assert renderer.stats()["shadow_packets"] == int(packets)
        ndiff = int((got != ref).sum())
        assert ndiff == 0, "%s build, yaw %g, LT_SHADOW_PACKETS=%s: %d of %d floats differ from the reference kernel" % (build, yaw, packets, ndiff, ref.size)
