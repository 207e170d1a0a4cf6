"""CPU tests of the oracle (oracle/lt_oracle.c): pinned against the reference's own known-answer and
invariance tests (/root/reference/tests/opencl_renderer_test.cc:51-228) on buffers dumped from the
reference's own host classes, and against the committed golden pixels."""
import os
import struct

import numpy as np
import pytest

from lens_trace_amd import scene as sc
from oracle import pyoracle as po
from tests.conftest import GOLDEN, golden_index

CAM = sc.camera_bytes(0.0, 2.5, -50.0, 0.0)   # Camera(0, 2.5, -50, 0) of every reference test


@pytest.fixture(scope="module")
def green_wall():
    return sc.load_ltsb(os.path.join(GOLDEN, "green_wall_O0.ltsb")).validate()


@pytest.fixture(scope="module")
def cornell():
    return sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()


def test_camera_buffer_matches_reference_dump(green_wall):
    # the reference's Camera(0,2.5,-50,0) buffer, dumped by its own class (src/camera.cpp:14-19)
    assert green_wall.camera == CAM == struct.pack("<6fI", 0, 2.5, -50, 0, 0, 0, 0)


def test_correct_color(green_wall):
    """RenderBufferTEST.CorrectColor (opencl_renderer_test.cc:185-228): floats x,x+1,x+2 for
    x = 0,24,...<10000 are (0,1,0)."""
    img = po.render(green_wall, CAM, 100, 100, po.BASIC, po.MODE_LINEAR).reshape(-1)
    for x in range(0, 100 * 100, 8 * 3):
        assert img[x + 0] == 0.0 and img[x + 1] == 1.0 and img[x + 2] == 0.0
    # stronger than the reference asserts: the wall fills the view, every pixel is (0,1,0)
    assert np.array_equal(img.reshape(-1, 3), np.tile(np.float32([0, 1, 0]), (10000, 1)))


@pytest.mark.parametrize("local", [(10, 10), (4, 25), (100, 1), (1, 1)])
def test_kernel_mode(green_wall, local):
    """RenderBufferTEST.KernelMode (:120-183): linearKernel == tileKernel at every 32nd float, with the
    MAX_FIT work block (min(maxWorkItemSizes, image) = 100x100, renderer_opencl.cpp:84-85)."""
    a = po.render_opencl_launch(green_wall, CAM, 100, 100, po.BASIC, po.MODE_LINEAR, (100, 100), local).reshape(-1)
    b = po.render_opencl_launch(green_wall, CAM, 100, 100, po.BASIC, po.MODE_TILE, (100, 100), local).reshape(-1)
    assert not np.isnan(a).any() and not np.isnan(b).any()
    assert np.array_equal(a[::32], b[::32])
    assert np.array_equal(a, b)


def test_custom_block_size(green_wall):
    """RenderBufferTEST.CustomBlockSize (:51-118): MAX_FIT vs CUSTOM work blocks 10x10 and 5x5 agree at
    every 32nd float of the first 10000."""
    outs = [po.render_opencl_launch(green_wall, CAM, 100, 100, po.BASIC, po.MODE_LINEAR, g, (1, 1)).reshape(-1)
            for g in [(100, 100), (10, 10), (5, 5)]]
    for x in range(0, 100 * 100, 32):
        assert outs[0][x] == outs[1][x] == outs[2][x]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])


def test_launch_decomposition_equals_whole_image(cornell):
    """Pixel value is a pure function of (x,y,W,H,camera,scene): the OpenCL work-block launch and the
    whole-image iteration (CUDA ceil-div semantics, renderer_cuda.cpp:74-88) agree wherever both write."""
    whole = po.render(cornell, CAM, 96, 64, po.ACCUMULATOR, po.MODE_LINEAR)
    blocks = po.render_opencl_launch(cornell, CAM, 96, 64, po.ACCUMULATOR, po.MODE_LINEAR, (32, 16), (8, 4))
    assert np.array_equal(whole, blocks)
    # truncating launch maths: 40x40 blocks on 96x64 reach only 80x40 pixels
    part = po.render_opencl_launch(cornell, CAM, 96, 64, po.ACCUMULATOR, po.MODE_LINEAR, (40, 40), (1, 1))
    assert np.array_equal(part[:40, :80], whole[:40, :80])
    assert np.isnan(part[40:]).all() and np.isnan(part[:, 80:]).all()


@pytest.mark.parametrize("row", golden_index(), ids=lambda r: r["tag"])
def test_golden_pixels_and_counters(row):
    s = sc.load_ltsb(os.path.join(GOLDEN, row["scene"] + ".ltsb")).validate()
    cam = sc.camera_bytes(0.0, 2.5, -50.0, row["yaw"], 0.0, 0.0, row["frame"])
    img, st = po.render(s, cam, row["W"], row["H"], po.PROGRAMS[row["program"]], row["mode"], threads=4, want_stats=True)
    want = np.load(os.path.join(GOLDEN, row["tag"] + ".npy"))
    assert np.array_equal(img, want)
    for k in ("rays", "shadow_rays", "node_visits", "tri_tests"):
        assert st[k] == row[k]


def test_tile_mode_is_unclamped_linear_is_clamped(cornell):
    """SURVEY Q14 (accumulator.cl:316-318 vs :356-358)."""
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0, 0, 7)
    lin = po.render(cornell, cam, 128, 128, po.ACCUMULATOR, po.MODE_LINEAR)
    til = po.render(cornell, cam, 128, 128, po.ACCUMULATOR, po.MODE_TILE)
    assert lin.min() >= 0.0 and lin.max() <= 1.0
    assert til.min() < 0.0                      # back-facing dot(toLight, normal) survives in tile mode
    assert np.array_equal(np.clip(til, 0.0, 1.0), lin)


def test_random_properties():
    """random() (accumulator.cl:63-66): in [0,1], quantised to float ulps of a (<= 2^-8 .. 2^-9 for
    large a), deterministic, and sensitive to the fused float dot."""
    rng = np.random.default_rng(1)
    for _ in range(2000):
        u, v = rng.uniform(-0.5, 0.5, 2).astype(np.float32)
        seed = float(rng.integers(0, 4096))
        r = po.random(u, v, seed)
        assert 0.0 <= r <= 1.0
        assert r == po.random(u, v, seed)
    # known values, computed independently in numpy with the same operation order
    for (u, v, seed) in [(0.25, -0.125, 3.0), (-0.5, -0.5, 0.0), (0.49609375, 0.0, 64.0)]:
        u32, v32 = np.float32(u), np.float32(v)
        prod = np.float64(u32) * np.float64(np.float32(12.9898))            # exact in double
        d = np.float32(np.float64(np.float32(prod)) + np.float64(v32) * np.float64(np.float32(78.233)))
        # fma(v, 78.233f, u*12.9898f): a single rounding of the exact sum (exact in double here)
        x = np.float64(d) + np.float64(1113.1) * np.float64(np.float32(seed))
        a = np.float32(np.sin(np.fmod(x, np.pi)) * 43758.5453)
        assert po.random(u, v, seed) == float(a - np.floor(a))


def _hand_scene(nodes, prims):
    mats = np.zeros(1, dtype=sc.MATERIAL_DTYPE)
    mats["diffuse"] = [0.25, 0.5, 0.75]
    mats["dissolve"] = 1.0
    mats["ior"] = 1.45
    lights = np.zeros(1, dtype=sc.LIGHT_DTYPE)
    return sc.Scene(nodes.view(np.uint8).reshape(-1), prims.view(np.uint8).reshape(-1), mats.view(np.uint8).reshape(-1),
                    lights.view(np.uint8).reshape(-1))


def test_multi_primitive_leaf_tests_only_first_triangle():
    """SURVEY Q2 (basic.cl:149-154): a leaf with primitiveCount 2 intersects primitives[offset] twice and
    never primitives[offset+1]."""
    prims = np.zeros(2, dtype=sc.PRIM_DTYPE)
    prims[0]["positionA"], prims[0]["positionB"], prims[0]["positionC"] = [-25, -25, 0], [25, -25, 0], [-25, 25, 0]
    prims[1]["positionA"], prims[1]["positionB"], prims[1]["positionC"] = [25, -25, 0], [25, 25, 0], [-25, 25, 0]
    for p in prims:
        p["normalA"] = p["normalB"] = p["normalC"] = [0, 0, 1]
    nodes = np.zeros(1, dtype=sc.NODE_DTYPE)
    nodes[0]["boundsMin"], nodes[0]["boundsMax"] = [-25, -25, -1e-6], [25, 25, 1e-6]
    nodes[0]["offset"], nodes[0]["primitiveCount"] = 0, 2
    s = _hand_scene(nodes, prims).validate()
    img, st = po.render(s, CAM, 64, 64, po.BASIC, want_stats=True)
    hit = img.sum(axis=2) > 0
    assert 0.2 < hit.mean() < 0.8                 # second triangle never hit: part of the wall is black
    assert st["tri_tests"] == 2 * 64 * 64         # the reference *calls* intersectTriangle twice per leaf
    # the same two triangles as two single-primitive leaves fill the view
    nodes3 = np.zeros(3, dtype=sc.NODE_DTYPE)
    nodes3["boundsMin"], nodes3["boundsMax"] = [-25, -25, -1e-6], [25, 25, 1e-6]
    nodes3[0]["offset"], nodes3[0]["axis"] = 2, 0
    nodes3[1]["offset"], nodes3[1]["primitiveCount"] = 0, 1
    nodes3[2]["offset"], nodes3[2]["primitiveCount"] = 1, 1
    full = po.render(_hand_scene(nodes3, prims).validate(), CAM, 64, 64, po.BASIC)
    assert (full.sum(axis=2) > 0).all()


def test_negative_t_hits_are_accepted():
    """intersectTriangle has no t > 0 test; a triangle behind the origin whose box still
    passes tMax > 0 wins with negative t."""
    prims = np.zeros(1, dtype=sc.PRIM_DTYPE)
    prims[0]["positionA"], prims[0]["positionB"], prims[0]["positionC"] = [-5, -5, -2], [5, -5, -2], [0, 5, -2]
    nodes = np.zeros(1, dtype=sc.NODE_DTYPE)
    nodes[0]["boundsMin"], nodes[0]["boundsMax"] = [-5, -5, -2], [5, 5, 3]   # box reaches in front of the origin
    nodes[0]["offset"], nodes[0]["primitiveCount"] = 0, 1
    s = _hand_scene(nodes, prims)
    hit, prim, tuv = po.trace(s, [0, 0, 0, 1], [0, 0, 1, 0])
    assert hit == 1 and prim == 0 and tuv[0] == -2.0


def test_running_mean_formula():
    """accumulator.frag:10-20: acc = (c + acc*n)/(n+1); frame 0 replaces."""
    rng = np.random.default_rng(0)
    frames = rng.uniform(0, 1, (5, 1000)).astype(np.float32)
    acc = np.full(1000, 123.0, dtype=np.float32)
    for n in range(5):
        po.accumulate(acc, frames[n], n)
    ref = frames[0].copy()
    for n in range(1, 5):
        ref = (frames[n] + ref * np.float32(n)) / np.float32(n + 1)
    assert np.array_equal(acc, ref)
    assert np.allclose(acc, frames.mean(axis=0), atol=1e-6)
