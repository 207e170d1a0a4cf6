"""GPU edge cases (-m gpu): the situations the reference's quirks and this backend's own
special paths (DEEP stack spill, non-finite rays, host-side validation) create, each against the CPU oracle bit for bit,
plus a seeded fuzz over random small scenes, cameras and programs."""
import os

import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd.renderer import KERNEL_MODE_TILE, RendererHIP
from oracle import pyoracle as po
from tests.conftest import fuzz_scene
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu

PATHS = {"basic": "basic.cl", "basic_lighting": "basic_lighting.cl", "accumulator": "accumulator.cl",
         "global_illumination": "examples/global_illumination/resources/kernels/global_illumination.cl",
         "global_illumination25": "resources/kernels/opencl/global_illumination.cl"}


@pytest.fixture(scope="module")
def renderer():
    r = RendererHIP(0)
    yield r
    r.close()


def both(renderer, s, prog, W, H, cam, mode=0, counters=True, **kw):
    got = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(PATHS[prog], (W, H, 3), got, s, pCamera=cam, kernelMode=mode, **kw))
    want = po.render(s, cam, W, H, po.PROGRAMS[prog], mode, gi_max_depth=kw.get("giMaxDepth", 0) or 16)
    assert np.array_equal(got, want), "%s: %d floats differ" % (prog, int((got != want).sum()))
    if counters:
        cnt = np.zeros((H, W, 4), dtype=np.float32)
        renderer.render(RenderPropertiesHIP(PATHS[prog], (W, H, 4), cnt, s, pCamera=cam, kernelMode=mode, pixelCounters=True, **kw))
        assert np.array_equal(cnt.astype(np.uint32), po.pixel_counters(s, cam, W, H, po.PROGRAMS[prog], mode, kw.get("giMaxDepth", 0) or 16))
    return got


def mats(colors, emissive=()):
    m = np.zeros(len(colors), dtype=sc.MATERIAL_DTYPE)
    m["ior"], m["dissolve"] = 1.45, 1.0
    for i, c in enumerate(colors):
        m[i]["diffuse"] = c
        if i in emissive:
            m[i]["emission"] = (1, 1, 1)
    return m


def lights_of(prims, materials):
    L = np.zeros(1, dtype=sc.LIGHT_DTYPE)
    idx = [i for i in range(len(prims)) if materials[prims[i]["materialIndex"]]["emission"].max() > 0][:64]
    L[0]["count"] = len(idx)
    L[0]["primitives"][:len(idx)] = idx
    return L


def scene_from(nodes, prims, materials):
    return sc.Scene(nodes.view(np.uint8).reshape(-1), prims.view(np.uint8).reshape(-1), materials.view(np.uint8).reshape(-1),
                    lights_of(prims, materials).view(np.uint8).reshape(-1)).validate()


def quad_prims(n, z0=0.0, dz=0.15, mat=lambda i: i % 3):
    """n triangles, alternately the two halves of a 10x10 wall quad, each on its own z plane in front of the camera."""
    p = np.zeros(n, dtype=sc.PRIM_DTYPE)
    for i in range(n):
        z = z0 - i * dz
        a, b, c = ([-5, -2.5, z], [5, -2.5, z], [-5, 7.5, z]) if i % 2 == 0 else ([5, -2.5, z], [5, 7.5, z], [-5, 7.5, z])
        p[i]["positionA"], p[i]["positionB"], p[i]["positionC"] = a, b, c
        p[i]["normalA"] = p[i]["normalB"] = p[i]["normalC"] = [0, 0, -1]
        p[i]["materialIndex"] = mat(i)
    return p


def caterpillar(prims):
    """A maximally unbalanced BVH: every interior node has one leaf child (left) and the rest of the chain (right);
    height = n - 1."""
    n = len(prims)
    lo = np.minimum(np.minimum(prims["positionA"], prims["positionB"]), prims["positionC"])
    hi = np.maximum(np.maximum(prims["positionA"], prims["positionB"]), prims["positionC"])
    nodes = np.zeros(2 * n - 1, dtype=sc.NODE_DTYPE)
    for k in range(n - 1):                      # interior node 2k covers prims k..n-1; its left child 2k+1 is leaf k
        nodes[2 * k]["boundsMin"], nodes[2 * k]["boundsMax"] = lo[k:].min(axis=0), hi[k:].max(axis=0)
        nodes[2 * k]["offset"], nodes[2 * k]["axis"] = 2 * k + 2, k % 3
        nodes[2 * k + 1]["boundsMin"], nodes[2 * k + 1]["boundsMax"] = lo[k], hi[k]
        nodes[2 * k + 1]["offset"], nodes[2 * k + 1]["primitiveCount"] = k, 1
    nodes[2 * n - 2]["boundsMin"], nodes[2 * n - 2]["boundsMax"] = lo[n - 1], hi[n - 1]
    nodes[2 * n - 2]["offset"], nodes[2 * n - 2]["primitiveCount"] = n - 1, 1
    return nodes


CAM = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 3)


@pytest.mark.parametrize("n", [30, 34, 50, 64])
def test_deep_bvh_uses_the_spilling_stack(renderer, n):
    """Heights 29 (LDS only), 33, 49, 63 (rows >= 32 spill to scratch: the DEEP instantiation)."""
    prims = quad_prims(n)
    m = mats([(0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8), (0.8, 0.8, 0.8)], emissive=(3,))
    prims[n // 2]["materialIndex"] = 3
    s = scene_from(caterpillar(prims), prims, m)
    for prog in ("basic", "accumulator", "global_illumination"):
        both(renderer, s, prog, 40, 24, CAM)


def test_a_counting_launch_with_fewer_lds_rows_than_the_tree_is_high_spills_and_finishes(renderer, monkeypatch):
    """LT_DEBUG_LDS_ROWS below the tree's height (round 2 hung a run that way: per-lane stack rows beyond the launch's LDS): the
    kernel is told how many rows it got and the host launches the form that keeps the rest of a lane's stack in private memory.
    One row for the Cornell box (height 9), pixels and per-pixel work counters against the oracle."""
    from tests.conftest import GOLDEN
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    for rows in ("1", "3", "40"):
        monkeypatch.setenv("LT_DEBUG_LDS_ROWS", rows)
        for prog in ("accumulator", "global_illumination"):
            both(renderer, s, prog, 48, 32, CAM)


def test_bvh_deeper_than_the_reference_stack_is_refused(renderer):
    prims = quad_prims(70)
    s = sc.Scene(caterpillar(prims).view(np.uint8).reshape(-1), prims.view(np.uint8).reshape(-1),
                 mats([(1, 1, 1)] * 3).view(np.uint8).reshape(-1), np.zeros(260, dtype=np.uint8))
    with pytest.raises(C.LensTraceError) as e:
        renderer.set_scene(s)
    assert e.value.code == C.LT_ERR_BAD_SCENE


def test_multi_primitive_leaves_first_triangle_only(renderer):
    """SURVEY Q2 on the device: primitiveCount 3 leaf -> only primitives[offset] is intersected, the work counter
    still counts 3 calls."""
    prims = quad_prims(6, dz=0.0)
    nodes = np.zeros(3, dtype=sc.NODE_DTYPE)
    nodes["boundsMin"], nodes["boundsMax"] = [-5, -2.5, -1e-3], [5, 7.5, 1e-3]
    nodes[0]["offset"], nodes[0]["axis"] = 2, 1
    nodes[1]["offset"], nodes[1]["primitiveCount"] = 0, 3
    nodes[2]["offset"], nodes[2]["primitiveCount"] = 2, 3      # first triangle of both leaves is a lower-left half
    s = scene_from(nodes, prims, mats([(0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8)]))
    img = both(renderer, s, "basic", 48, 48, CAM)
    assert 0.1 < (img.sum(axis=2) > 0).mean() < 0.9


def test_emissive_primitive_zero_makes_every_miss_white(renderer):
    """the hit-a-light test compares primitiveIndex without looking at hitType; a miss has index 0."""
    prims = quad_prims(2, dz=0.0)
    prims["positionA"] *= 0.2; prims["positionB"] *= 0.2; prims["positionC"] *= 0.2     # small wall: most rays miss
    prims[0]["materialIndex"], prims[1]["materialIndex"] = 1, 0
    nodes = caterpillar(prims)
    s = scene_from(nodes, prims, mats([(0.5, 0.5, 0.5), (0.8, 0.8, 0.8)], emissive=(1,)))
    assert s.light_view[0]["primitives"][0] == 0
    img = both(renderer, s, "accumulator", 40, 40, CAM)
    assert (img == 1.0).all(axis=2).mean() > 0.5
    both(renderer, s, "global_illumination", 40, 40, CAM)


def test_scene_without_lights(renderer):
    prims = quad_prims(4)
    s = scene_from(caterpillar(prims), prims, mats([(0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8)]))
    assert s.light_view[0]["count"] == 0
    for prog in ("accumulator", "basic_lighting", "global_illumination"):
        both(renderer, s, prog, 32, 20, CAM)


def test_axis_parallel_and_degenerate_rays(renderer):
    """Image-centre column and row (direction components exactly 0 -> invDir = inf -> NaN box tests), a camera standing in
    the plane of a box face, degenerate (zero-area) triangles."""
    prims = quad_prims(8, z0=0.0, dz=0.5)
    prims[3]["positionB"] = prims[3]["positionA"]            # zero-area triangle: det = 0, rejected by the epsilon test
    m = mats([(0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8), (1, 1, 1)], emissive=(3,))
    prims[5]["materialIndex"] = 3
    from lens_trace_amd.scene import build_from_triangles
    s = build_from_triangles(np.stack([prims["positionA"], prims["positionB"], prims["positionC"]], axis=1),
                             np.stack([prims["normalA"], prims["normalB"], prims["normalC"]], axis=1), prims["materialIndex"], m)
    for cam in (sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0, 0, 1),        # centre ray hits x = 0, y = 2.5 box planes
                sc.camera_bytes(5.0, 7.5, -50.0, 0.0, 0, 0, 2),        # film centre on the scene's max corner
                sc.camera_bytes(-5.0, -2.5, -20.0, 0.0, 0, 0, 2)):
        for prog in ("basic", "accumulator", "global_illumination"):
            both(renderer, s, prog, 33, 33, cam)                       # odd size: a pixel exactly at the film centre? (16/33 no)
            both(renderer, s, prog, 32, 32, cam)                       # even size: x = 16 -> film.x = 0 exactly


@pytest.mark.parametrize("seed", range(int(os.environ.get("LT_FUZZ_SEEDS", "12"))))   # LT_FUZZ_SEEDS=300 for a long soak
def test_fuzz_random_scenes(renderer, monkeypatch, seed):
    monkeypatch.setenv("LT_GI_MEGAKERNEL", str(seed % 2))       # alternate the two GI execution paths
    s, cam, W, H, rng = fuzz_scene(seed)
    for prog in ("basic", "accumulator", "global_illumination"):
        both(renderer, s, prog, W, H, cam, mode=int(rng.integers(0, 2)), counters=(seed % 3 == 0))
    if seed % 4 == 0:
        both(renderer, s, "basic_lighting", min(W, 24), min(H, 16), cam, counters=False)
        both(renderer, s, "global_illumination25", min(W, 16), min(H, 12), cam, counters=False, giMaxDepth=5)


@pytest.mark.parametrize("name", ["wall", "soup", "blob"])
def test_sah_built_scenes_match_the_oracle_bit_for_bit(renderer, monkeypatch, name):
    """Scenes built with ACCELERATION_STRUCTURE_TYPE_BVH_SAH (same layouts, taller and unbalanced trees: other LDS stack
    heights, other near / far patterns) through both shadow-ray walks and both GI paths, against the CPU oracle on the same
    buffers: pixels and per-pixel work counters."""
    from lens_trace_amd import synth
    s = {"wall": lambda: synth.heightfield_wall(96, bvh=sc.BVH_SAH), "soup": lambda: synth.triangle_soup(30000, bvh=sc.BVH_SAH),
         "blob": lambda: synth.blob_in_box(4, bvh=sc.BVH_SAH)}[name]().validate()
    W, H = 160, 96
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.02 if name == "soup" else 0.0, 0.0, 0.0, 3)
    want, st = po.render(s, cam, W, H, po.ACCUMULATOR, threads=8, want_stats=True)
    for packets in ("0", "1", "3"):
        monkeypatch.setenv("LT_SHADOW_PACKETS", packets)
        got = np.empty((H, W, 3), dtype=np.float32)
        renderer.render(RenderPropertiesHIP(PATHS["accumulator"], (W, H, 3), got, s, pCamera=cam))
        assert np.array_equal(got, want), "LT_SHADOW_PACKETS=%s" % packets
    renderer.render(RenderPropertiesHIP(PATHS["accumulator"], (W, H, 3), got, s, pCamera=cam, collectStats=True))
    hs = renderer.stats()
    for k in ("rays", "shadow_rays", "node_visits", "tri_tests"):
        assert hs[k] == st[k], k
    want = po.render(s, cam, 64, 40, po.GI, threads=8)
    for mega in ("0", "1"):
        monkeypatch.setenv("LT_GI_MEGAKERNEL", mega)
        got = np.empty((40, 64, 3), dtype=np.float32)
        renderer.render(RenderPropertiesHIP(PATHS["global_illumination"], (64, 40, 3), got, s, pCamera=cam))
        assert np.array_equal(got, want), "LT_GI_MEGAKERNEL=%s" % mega
