"""GPU tests (-m gpu) of the other BASELINE.json configurations at their stated sizes (they are parity cases, not bench
lines): config 2 (Cornell box, global_illumination, 1920x1080, depth 16 and 4), config 3 (~70 k-triangle blob in a box,
accumulator, 1080p, 64 frames), config 5 (~250 k-triangle colonnade, 4K, 256 progressive frames).  Whole frames are checked
through size-independent properties; sampled rows are compared with the CPU oracle bit for bit."""
import os
import time

import numpy as np
import pytest

from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from lens_trace_amd.renderer import RendererHIP
from oracle import pyoracle as po
from tests.conftest import GOLDEN
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu
ACC = "examples/accumulator/resources/kernels/accumulator.cl"
GI = "examples/global_illumination/resources/kernels/global_illumination.cl"


@pytest.fixture(scope="module")
def renderer():
    r = RendererHIP(0)
    yield r
    r.close()


def render(renderer, scene, path, W, H, frame=None, **kw):
    out = np.empty((H, W, 3), dtype=np.float32)
    cam = scene.camera if frame is None else sc.camera_with_frame(scene.camera, frame)
    t0 = time.perf_counter()
    renderer.render(RenderPropertiesHIP(path, (W, H, 3), out, scene, pCamera=cam, **kw))
    return out, time.perf_counter() - t0


def rows_equal_oracle(got, scene, cam, W, H, program, rows, **kw):
    for y in rows:
        want = po.render(scene, cam, W, H, program, rows=(y, y + 1), threads=1, **kw)
        assert np.array_equal(got[y], want[y]), "row %d" % y


def test_config2_cornell_global_illumination_1080p(renderer):
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    W, H = 1920, 1080
    for depth in (16, 4):                       # the reference constant is 16; BASELINE's perf configuration says 4 bounces
        got, dt = render(renderer, s, GI, W, H, frame=2, giMaxDepth=depth, collectStats=True)
        st = renderer.stats()
        print("config 2: depth %d, %.1f ms kernel, %.0f Mrays/s, %.2f rays/pixel" % (depth, st["kernel_ms"], st["rays"] / st["kernel_ms"] / 1e3, st["rays"] / (W * H)))
        rows_equal_oracle(got, s, sc.camera_with_frame(s.camera, 2), W, H, po.GI, (0, 411, 540, 541, 799, 1079), gi_max_depth=depth)
        assert got.min() >= 0.0 and got.max() <= 1.0
    # 35 % of the frame shows the box, the rest misses (primitive 0 is not a light here)
    assert 0.2 < (got.sum(axis=2) > 0).mean() < 0.6
    # 16 frames per call (the wavefront pipeline carries all of them through one set of stage launches at 16 bounces, the
    # single kernel renders them in one launch at 4) == folding 16 single-frame calls with accumulator.frag's formula
    for depth in (16, 4):
        many, _ = render(renderer, s, GI, W, H, frameFirst=1, frameCount=16, accumulate=True, giMaxDepth=depth)
        st = renderer.stats()
        print("config 2: depth %d, 16 frames per call: %.2f ms kernel per frame, %d launches" % (depth, st["kernel_ms"] / 16, st["kernel_launches"]))
        acc = np.zeros((H, W, 3), dtype=np.float32)
        for i, f in enumerate(range(1, 17)):
            frame, _ = render(renderer, s, GI, W, H, frame=f, giMaxDepth=depth)
            po.accumulate(acc.reshape(-1), frame.reshape(-1), i)
        assert np.array_equal(many, acc)


def test_config2_full_frame_matches_reference_gi_kernel(renderer, monkeypatch):
    """BASELINE config 2 at its stated size against the reference's own global_illumination.cl
    (examples/global_illumination/resources/kernels/global_illumination.cl:241-375; its bounce count is the constant 16)
    compiled for gfx950 with the reference's NULL build options: the default flavour, both GI execution paths, bit for bit over
    the 1920x1080 frame; and the strict flavour against the strict build of the same file."""
    from lens_trace_amd.renderer import RenderPropertiesHIP as DefaultFlavourProps
    from oracle import ref_gpu
    if not ref_gpu.available("global_illumination", "default"):
        pytest.skip("oracle/_ref/global_illumination.default.co not built (needs /root/reference at build time)")
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    W, H = 1920, 1080
    cam = sc.camera_with_frame(s.camera, 2)
    for build in ("default", "strict"):
        ref = ref_gpu.render(s, cam, W, H, "global_illumination", build)
        for mega in ("0", "1"):
            monkeypatch.setenv("LT_GI_MEGAKERNEL", mega)
            got = np.empty((H, W, 3), dtype=np.float32)
            renderer.render(DefaultFlavourProps(GI, (W, H, 3), got, s, pCamera=cam, strictMath=(build == "strict")))
            ndiff = int((got != ref).sum())
            assert ndiff == 0, "%s build, LT_GI_MEGAKERNEL=%s: %d of %d floats differ from the reference kernel" % (build, mega, ndiff, ref.size)
    # and the running mean of 4 frames through the fused pipeline == the reference's frames folded in float32
    acc = None
    for k in range(4):
        c = ref_gpu.render(s, sc.camera_with_frame(s.camera, 1 + k), W, H, "global_illumination", "default")
        acc = c if k == 0 else ((c + acc * np.float32(k)) / np.float32(k + 1)).astype(np.float32)
    monkeypatch.delenv("LT_GI_MEGAKERNEL")
    got = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(DefaultFlavourProps(GI, (W, H, 3), got, s, pCamera=s.camera, frameFirst=1, frameCount=4, accumulate=True))
    assert int((got != acc).sum()) == 0


def test_config3_blob_accumulator_1080p_64_frames(renderer):
    s = synth.blob_in_box().validate()
    assert 60000 < s.n_prims < 80000
    W, H = 1920, 1080
    got, dt = render(renderer, s, ACC, W, H, frameFirst=1, frameCount=64, accumulate=True)
    st = renderer.stats()
    print("config 3: %d triangles, 64 frames in %.1f ms kernel time" % (s.n_prims, st["kernel_ms"]))
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for i, f in enumerate(range(1, 65)):
        frame, _ = render(renderer, s, ACC, W, H, frame=f)
        if f in (1, 64):
            rows_equal_oracle(frame, s, sc.camera_with_frame(s.camera, f), W, H, po.ACCUMULATOR, (3, 540, 1000))
        po.accumulate(acc.reshape(-1), frame.reshape(-1), i)
    assert np.array_equal(got, acc)
    # progressive refinement converges: the 64-frame mean is closer to the 32+32 halves' mean than a single frame is
    assert got.min() >= 0.0 and got.max() <= 1.0


def test_config5_colonnade_4k_256_progressive_frames(renderer):
    s = synth.colonnade().validate()
    assert 230000 < s.n_prims < 280000
    W, H = 3840, 2160
    whole, dt = render(renderer, s, ACC, W, H, frameFirst=1, frameCount=256, accumulate=True)
    st = renderer.stats()
    print("config 5: %d triangles, 256 frames at 4K: %.0f ms kernel time (%.2f ms / frame)" % (s.n_prims, st["kernel_ms"], st["kernel_ms"] / 256))
    # the same 256 frames as four calls of 64 that continue the caller's accumulator (accumulate_base)
    part = np.zeros((H, W, 3), dtype=np.float32)
    for k in range(4):
        renderer.render(RenderPropertiesHIP(ACC, (W, H, 3), part, s, pCamera=s.camera, frameFirst=1 + 64 * k, frameCount=64,
                                            accumulate=True, accumulateBase=64 * k))
    assert np.array_equal(part, whole)
    assert whole.min() >= 0.0 and whole.max() <= 1.0
    single, _ = render(renderer, s, ACC, W, H, frame=200)
    rows_equal_oracle(single, s, sc.camera_with_frame(s.camera, 200), W, H, po.ACCUMULATOR, (10, 1080, 2000))
    # a 256-sample mean is smoother than one sample: smaller mean absolute horizontal gradient
    assert np.abs(np.diff(whole, axis=1)).mean() < np.abs(np.diff(single, axis=1)).mean()


# ---- the configurations at their stated sizes against the reference's OWN kernels, default flavour ----
def _default_props():
    from lens_trace_amd.renderer import RenderPropertiesHIP as DefaultFlavourProps
    return DefaultFlavourProps


def _need_ref(kernel):
    from oracle import ref_gpu
    if not ref_gpu.available(kernel, "default"):
        pytest.skip("oracle/_ref/%s.default.co not built (needs /root/reference at build time)" % kernel)
    return ref_gpu


def test_config1_cornell_basic_512_matches_the_reference_kernel(renderer):
    """BASELINE config 1 at its stated size (Cornell box, primary rays only, 512x512) against basic.cl
    (resources/kernels/opencl/basic.cl:279-342) as the reference builds it, and against the CPU oracle in the portable flavour."""
    ref_gpu = _need_ref("basic")
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    W = H = 512
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 0)
    got = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(_default_props()("basic.cl", (W, H, 3), got, s, pCamera=cam))
    assert int((got != ref_gpu.render(s, cam, W, H, "basic", "default")).sum()) == 0
    portable = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP("basic.cl", (W, H, 3), portable, s, pCamera=cam))
    assert np.array_equal(portable, po.render(s, cam, W, H, po.BASIC, threads=8))


def test_config3_blob_1080p_matches_the_reference_accumulator_kernel(renderer):
    """BASELINE config 3's scene and size (69 950-triangle blob in a box, 1920x1080) against accumulator.cl
    (examples/accumulator/resources/kernels/accumulator.cl:132-217, :219-318) as the reference builds it: one whole frame, and the
    running mean of four through the fused launch against the reference's frames folded in float32."""
    ref_gpu = _need_ref("accumulator")
    s = synth.blob_in_box().validate()
    W, H = 1920, 1080
    Props = _default_props()
    acc = None
    for k in range(4):
        cam = sc.camera_with_frame(s.camera, 1 + k)
        c = ref_gpu.render(s, cam, W, H, "accumulator", "default")
        if k == 0:
            got = np.empty((H, W, 3), dtype=np.float32)
            renderer.render(Props(ACC, (W, H, 3), got, s, pCamera=cam))
            assert int((got != c).sum()) == 0
        acc = c if k == 0 else ((c + acc * np.float32(k)) / np.float32(k + 1)).astype(np.float32)
    got = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(Props(ACC, (W, H, 3), got, s, pCamera=s.camera, frameFirst=1, frameCount=4, accumulate=True))
    assert int((got != acc).sum()) == 0


def test_config5_colonnade_4k_matches_the_reference_accumulator_kernel(renderer, monkeypatch):
    """BASELINE config 5's scene and size (255 k-triangle colonnade, 3840x2160), one frame, against accumulator.cl as the reference
    builds it -- with the shadow rays through the walk the library picks and through the queue (lt_trace_kernel)."""
    ref_gpu = _need_ref("accumulator")
    s = synth.colonnade().validate()
    W, H = 3840, 2160
    cam = sc.camera_with_frame(s.camera, 5)
    ref = ref_gpu.render(s, cam, W, H, "accumulator", "default")
    for forced in (None, "3"):
        if forced:
            monkeypatch.setenv("LT_SHADOW_PACKETS", forced)
        got = np.empty((H, W, 3), dtype=np.float32)
        renderer.render(_default_props()(ACC, (W, H, 3), got, s, pCamera=cam))
        assert int((got != ref).sum()) == 0, "LT_SHADOW_PACKETS=%s" % forced


def test_wall_1m_triangles_global_illumination_1080p_matches_the_reference_kernel(renderer, monkeypatch):
    """The 1 002 530-triangle wall at 1920x1080 through global_illumination.cl
    (examples/global_illumination/resources/kernels/global_illumination.cl:241-375, 16 bounces) as the reference builds it: the
    wavefront pipeline -- extension and shadow rays through lt_trace_kernel over the 4-wide groups of the own hierarchy, equal-t
    ties by the reference's leaf order on a 22-level tree -- its one-kernel bounce stage, and the single kernel."""
    ref_gpu = _need_ref("global_illumination")
    s = synth.heightfield_wall(708).validate()
    W, H = 1920, 1080
    cam = sc.camera_with_frame(s.camera, 2)
    ref = ref_gpu.render(s, cam, W, H, "global_illumination", "default")
    for env in ({"LT_GI_MEGAKERNEL": "0"}, {"LT_GI_MEGAKERNEL": "0", "LT_GI_TRACE": "0"}, {"LT_GI_MEGAKERNEL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = np.empty((H, W, 3), dtype=np.float32)
        renderer.render(_default_props()(GI, (W, H, 3), got, s, pCamera=cam))
        ndiff = int((got != ref).sum())
        assert ndiff == 0, "%s: %d of %d floats differ from the reference kernel" % (env, ndiff, ref.size)
        monkeypatch.delenv("LT_GI_TRACE", raising=False)
