"""GPU (-m gpu): the backend's own hierarchy over the caller's leaves (lens_trace_amd/csrc/lt_retree.hpp) never changes a pixel.

* bit-equal hits -- coincident triangles with different materials, whose accepted hits have the same t to the last bit -- are
  given to the triangle the REFERENCE's depth-first order meets first (intersectTriangle accepts `t < payload.t`, acc.cl:104),
  whatever order the backend's walks meet them in (SceneDev::rank8), for camera rays, lens rays and GI bounce rays;
* the caller's splits (LT_RETREE=0), the backend's own (default) and the reference-order walks of the counting kernels give
  the same image, equal to the CPU oracle's;
* a scene whose boxes do not nest gets no hierarchy of the backend's own (every ray then walks the caller's tree in the
  reference's order) and still equals the oracle."""
import numpy as np
import pytest

from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from lens_trace_amd.renderer import RendererHIP
from oracle import pyoracle as po
from tests.conftest import oracle_props as RenderPropertiesHIP

pytestmark = pytest.mark.gpu

PATHS = {"basic": "basic.cl", "accumulator": "accumulator.cl",
         "global_illumination": "examples/global_illumination/resources/kernels/global_illumination.cl"}


def fresh(monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return RendererHIP(0)


def render(r, s, prog, W, H, cam, **kw):
    out = np.empty((H, W, 3), dtype=np.float32)
    r.render(RenderPropertiesHIP(PATHS[prog], (W, H, 3), out, s, pCamera=cam, **kw))
    return out


def doubled_scene(seed, lens=False):
    """Random triangles, every one of them present TWICE with different materials (so each hit comes as a bit-equal pair and the
    colour tells which copy won), the copies shuffled apart in the input so the BVH puts them in either order."""
    rng = np.random.default_rng(seed)
    n = 120
    centre = np.stack([rng.uniform(-4, 4, n), rng.uniform(-1.5, 6.5, n), rng.uniform(-4, 0, n)], axis=-1)
    pos = (centre[:, None, :] + rng.normal(0, 0.9, (n, 3, 3))).astype(np.float32)
    nrm = np.tile(np.float32([0, 0, -1]), (n, 3, 1))
    pos2, nrm2 = np.concatenate([pos, pos]), np.concatenate([nrm, nrm])
    k = 6
    m = np.zeros(k, dtype=sc.MATERIAL_DTYPE)
    m["diffuse"] = rng.uniform(0.1, 1, (k, 3))
    m["ior"], m["dissolve"] = 1.3, 1.0
    if lens:
        m["dissolve"][1] = 0.25
    m[k - 1]["emission"] = (1, 1, 1)
    mi = np.concatenate([rng.integers(0, 3, n), rng.integers(3, k, n)]).astype(np.int32)
    perm = rng.permutation(2 * n)
    return sc.build_from_triangles(pos2[perm], nrm2[perm], mi[perm], m).validate()


@pytest.mark.parametrize("seed", range(4))
def test_bit_equal_hits_go_to_the_references_first_leaf(monkeypatch, seed):
    s = doubled_scene(seed, lens=(seed % 2 == 1))
    r = fresh(monkeypatch)
    W, H = 96, 64
    for yaw, frame in ((0.0, 0), (0.05, 3)):
        cam = sc.camera_bytes(0.3, 2.5, -50.0, yaw, 0.0, 0.0, frame)
        for prog in ("basic", "accumulator", "global_illumination"):
            for mega in ("0", "1"):
                monkeypatch.setenv("LT_GI_MEGAKERNEL", mega)
                got = render(r, s, prog, W, H, cam)
                want = po.render(s, cam, W, H, po.PROGRAMS[prog])
                assert r.stats()["own_tree_height"] > 0
                assert np.array_equal(got, want), "%s seed %d yaw %g: %d floats differ" % (prog, seed, yaw, int((got != want).sum()))
    r.close()


@pytest.mark.parametrize("name", ["wall", "soup", "blob"])
def test_callers_splits_and_own_splits_give_the_same_frame(monkeypatch, name):
    s = {"wall": lambda: synth.heightfield_wall(96), "soup": lambda: synth.triangle_soup(20000), "blob": lambda: synth.blob_in_box(4)}[name]().validate()
    W, H = 200, 120
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.01, 0.0, 0.0, 2)
    want = po.render(s, cam, W, H, po.ACCUMULATOR, threads=8)
    frames = {}
    for retree in ("1", "0"):
        r = fresh(monkeypatch, LT_RETREE=retree)
        for packets in ("0", "1", "2", "3"):
            monkeypatch.setenv("LT_SHADOW_PACKETS", packets)
            frames[retree, packets] = render(r, s, "accumulator", W, H, cam)
            assert r.stats()["own_tree_height"] > 0
        gi = render(r, s, "global_illumination", 64, 48, cam, giMaxDepth=4)
        assert np.array_equal(gi, po.render(s, cam, 64, 48, po.PROGRAMS["global_illumination"], gi_max_depth=4, threads=8))
        r.close()
    for key, f in frames.items():
        assert np.array_equal(f, want), key


def test_a_scene_whose_boxes_do_not_nest_walks_the_callers_tree(monkeypatch):
    s = synth.blob_in_box(3).validate()
    nodes = s.node_view
    leaves = np.flatnonzero(nodes["primitiveCount"] != 0)
    for k in leaves[::7]:                     # leaves that poke out of their ancestors: legal for the reference's traversal
        nodes["boundsMax"][k] += np.float32(0.75)
        nodes["boundsMin"][k] -= np.float32(0.25)
    W, H = 128, 96
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, 1)
    r = fresh(monkeypatch)
    for prog in ("basic", "accumulator", "global_illumination"):
        got = render(r, s, prog, W, H, cam)
        assert r.stats()["own_tree_height"] == -1
        want = po.render(s, cam, W, H, po.PROGRAMS[prog], threads=8)
        assert np.array_equal(got, want), prog
    r.close()
