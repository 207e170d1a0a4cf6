"""GPU (-m gpu): scene preparation on the device (lens_trace_amd/csrc/lt_prep.hip) against the host's (lt_retree.hpp).

lt_hip_set_scene derives four things from the caller's node buffer: the structural verdict, the leaf order table of the
reference's walk, the backend's own binned-SAH hierarchy over the caller's leaves, and its collapse into 4-wide groups.  The
device path makes them with kernels; the host path (LT_DEVICE_BUILD=0, and what small or unusual scenes get) with threads.
Both are the same function of the buffer: these tests read the structures back (lt_hip_read_scene_structure) and compare them
byte for byte -- the own tree node for node, the table, every 64-byte record of the per-lane walks -- over the synthetic scenes,
random triangle sets, degenerate ones (coincident centroids, coincident triangles), both split rules (LT_RETREE=0 / 1) and a
height limit with no slack, where the median rule takes over from the planes.  Buffers the device path declines (a leaf shared
by two parents, a box outside its parent's, unreachable nodes, an index out of range) come out as they always did."""
import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from lens_trace_amd.renderer import RendererHIP
from tests.conftest import oracle_props as RenderPropertiesHIP

pytestmark = pytest.mark.gpu


def structures(monkeypatch, scene, device, **env):
    monkeypatch.setenv("LT_DEVICE_BUILD", "1" if device else "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = RendererHIP(0)
    r.set_scene(scene)
    got = [r.scene_structure(k) for k in range(4)]
    r.close()
    for k in env:
        monkeypatch.delenv(k)
    return got


def same(monkeypatch, scene, **env):
    host = structures(monkeypatch, scene, False, **env)
    dev = structures(monkeypatch, scene, True, **env)
    assert host[3][3] == 0 and dev[3][3] == 1, (host[3], dev[3])
    assert host[3][:3] == dev[3][:3], (host[3], dev[3])
    assert host[0] is not None and dev[0] is not None
    a, b = host[0].view(np.uint8).reshape(-1, 32), dev[0].view(np.uint8).reshape(-1, 32)
    assert a.shape == b.shape
    bad = np.flatnonzero((a != b).any(axis=1))
    assert len(bad) == 0, "own tree: %d of %d nodes differ, first at %d: host %s device %s" % (len(bad), len(a), bad[0], host[0][bad[0]], dev[0][bad[0]])
    assert np.array_equal(host[1], dev[1]), "leaf order table differs in %d entries" % int((host[1] != dev[1]).sum())
    assert np.array_equal(host[2], dev[2]), "per-lane walk records differ in %d bytes" % int((host[2] != dev[2]).sum())
    return dev


def random_triangles(seed, n, spread=4.0, size=0.3, coincident=0):
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-spread, spread, (n, 3))
    if coincident:   # groups of triangles that share one centroid exactly (symmetric about it), or are the same triangle
        centre[: coincident] = centre[0]
    pos = (centre[:, None, :] + rng.normal(0, size, (n, 3, 3))).astype(np.float32)
    if coincident:
        pos[coincident // 2: coincident] = pos[coincident // 2]
    nrm = np.tile(np.float32([0, 0, -1]), (n, 3, 1))
    m = np.zeros(3, dtype=sc.MATERIAL_DTYPE)
    m["diffuse"] = rng.uniform(0.1, 1, (3, 3))
    m["ior"], m["dissolve"] = 1.3, 1.0
    m[2]["emission"] = (1, 1, 1)
    mi = rng.integers(0, 2, n).astype(np.int32)
    mi[0] = 2   # one emissive triangle
    return sc.build_from_triangles(pos, nrm, mi, m).validate()


SCENES = {
    "wall": lambda: synth.heightfield_wall(96),
    "soup": lambda: synth.triangle_soup(30000),
    "blob": lambda: synth.blob_in_box(5),
    "colonnade": lambda: synth.colonnade(6, 32, 12),
    "mixed": lambda: synth.wall_and_soup(60, 9000),
    "cornell": lambda: sc.load_ltsb("tests/golden/cornell_box_O0.ltsb"),
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_device_and_host_preparation_agree_on_the_synthetic_scenes(monkeypatch, name):
    s = SCENES[name]().validate()
    dev = same(monkeypatch, s)
    assert dev[3][0] > 0
    same(monkeypatch, s, LT_RETREE="0")


@pytest.mark.parametrize("n", [2, 3, 5, 64, 65, 130, 2048, 2049, 2100, 5000, 70000])
def test_every_range_size_takes_its_path(monkeypatch, n):
    """2..64 leaves: one wavefront finishes the subtree; up to 2048: one wavefront per range, level by level; above: many workgroups."""
    s = random_triangles(n, n)
    same(monkeypatch, s)
    same(monkeypatch, s, LT_RETREE_SLACK="0")


@pytest.mark.parametrize("seed,n,coincident", [(1, 300, 40), (2, 3000, 700), (3, 9000, 6000), (4, 70, 70), (5, 4500, 4500)])
def test_coincident_centroids_and_triangles(monkeypatch, seed, n, coincident):
    s = random_triangles(seed, n, coincident=coincident)
    same(monkeypatch, s)
    same(monkeypatch, s, LT_RETREE_SLACK="0")
    same(monkeypatch, s, LT_RETREE_SLACK="1")


@pytest.mark.parametrize("split", [sc.BVH_MEDIAN, sc.BVH_SAH])
def test_the_bench_scene_at_a_quarter_of_its_size(monkeypatch, split):
    s = synth.heightfield_wall(354, bvh=split).validate()
    dev = same(monkeypatch, s)
    assert dev[3][0] >= 18


def test_buffers_the_device_declines_keep_their_host_verdicts(monkeypatch):
    monkeypatch.setenv("LT_DEVICE_BUILD", "1")
    base = random_triangles(9, 400)
    r = RendererHIP(0)

    def with_nodes(nodes, prims=None):
        s = sc.Scene(nodes=np.ascontiguousarray(nodes).view(np.uint8).reshape(-1), prims=base.prims if prims is None else prims.view(np.uint8).reshape(-1),
                     materials=base.materials, lights=base.lights, camera=base.camera)
        r.set_scene(s)
        return r.scene_structure(3)

    good = base.node_view.copy()
    info = with_nodes(good.copy())
    assert info[3] == 1 and info[0] > 0
    leaves = np.flatnonzero(good["primitiveCount"] != 0)
    interior = np.flatnonzero(good["primitiveCount"] == 0)
    # a box outside its parent's: no hierarchy of the backend's own (the scene walks the caller's tree), prepared by the host
    n = good.copy()
    n["boundsMax"][leaves[5]] += 100.0
    info = with_nodes(n)
    assert info[3] == 0 and r.scene_structure(0) is None
    # two leaves on one primitive: the own tree without 4-wide groups is not kept either
    n = good.copy()
    n["offset"][leaves[7]] = n["offset"][leaves[3]]
    info = with_nodes(n)
    assert info[3] == 0
    # an unreachable tail (the reference's traversal never gets there): harmless, host
    n = np.concatenate([good, good[-2:]])
    info = with_nodes(n)
    assert info[3] == 0 and info[0] > 0
    # malformed: the host words the error
    n = good.copy()
    n["offset"][interior[3]] = len(n) + 5
    with pytest.raises(C.LensTraceError, match="children out of range"):
        with_nodes(n)
    n = good.copy()
    n["offset"][leaves[2]] = base.n_prims + 1
    with pytest.raises(C.LensTraceError, match="primitivesOffset out of range"):
        with_nodes(n)
    p = base.prim_view.copy()
    p["materialIndex"][11] = 77
    with pytest.raises(C.LensTraceError, match="materialIndex out of range"):
        with_nodes(good.copy(), p)
    # ... and a good scene afterwards is prepared on the device again
    assert with_nodes(good.copy())[3] == 1
    r.close()


def test_frames_do_not_depend_on_who_prepared_the_scene(monkeypatch):
    s = synth.wall_and_soup(60, 9000).validate()
    cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.02, 0.0, 0.0, 1)
    frames = []
    for device in ("0", "1"):
        monkeypatch.setenv("LT_DEVICE_BUILD", device)
        r = RendererHIP(0)
        for prog in ("accumulator.cl", "examples/global_illumination/resources/kernels/global_illumination.cl"):
            out = np.empty((90, 160, 3), dtype=np.float32)
            r.render(RenderPropertiesHIP(prog, (160, 90, 3), out, s, pCamera=cam))
            frames.append(out)
        assert r.scene_structure(3)[3] == int(device)
        r.close()
    assert np.array_equal(frames[0], frames[2]) and np.array_equal(frames[1], frames[3])


def chain_scene(n, seed=3):
    """The caller's tree as a right-deep chain (interior k: left = leaf k, right = the rest): height n - 1, proper pre-order."""
    base = random_triangles(seed, n)
    pv = base.prim_view
    lo = np.minimum(np.minimum(pv["positionA"], pv["positionB"]), pv["positionC"])
    hi = np.maximum(np.maximum(pv["positionA"], pv["positionB"]), pv["positionC"])
    nodes = np.zeros(2 * n - 1, dtype=sc.NODE_DTYPE)
    for k in range(n - 1):
        i = 2 * k
        nodes[i]["boundsMin"], nodes[i]["boundsMax"] = lo[k:].min(axis=0), hi[k:].max(axis=0)
        nodes[i]["offset"], nodes[i]["primitiveCount"], nodes[i]["axis"] = i + 2, 0, k % 3
        nodes[i + 1]["boundsMin"], nodes[i + 1]["boundsMax"] = lo[k], hi[k]
        nodes[i + 1]["offset"], nodes[i + 1]["primitiveCount"] = k, 1
    nodes[-1]["boundsMin"], nodes[-1]["boundsMax"] = lo[n - 1], hi[n - 1]
    nodes[-1]["offset"], nodes[-1]["primitiveCount"] = n - 1, 1
    return sc.Scene(nodes=nodes.view(np.uint8).reshape(-1), prims=base.prims, materials=base.materials, lights=base.lights, camera=base.camera)


def test_a_deep_callers_tree(monkeypatch):
    """A chain of 40 leaves (height 39: the walks up the tree take that many steps) is prepared on the device like on the host; a
    chain of 70 (deeper than the reference's 64-entry stack) is refused in the host's words."""
    same(monkeypatch, chain_scene(40))
    monkeypatch.setenv("LT_DEVICE_BUILD", "1")
    r = RendererHIP(0)
    with pytest.raises(C.LensTraceError, match="deeper than the reference's 64-entry traversal stack"):
        r.set_scene(chain_scene(70))
    r.set_scene(chain_scene(64))
    assert r.scene_structure(3)[3] == 1 and r.stats is not None
    r.close()
