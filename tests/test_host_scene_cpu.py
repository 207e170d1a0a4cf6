"""CPU tests of the host-side scene classes (liblenstrace.so: own .obj/.mtl reader + deterministic BVH builder,
lens_trace_amd/host/scene_host.cpp), the SURVEY section 8(f)-1 row.  The buffers they emit are checked against the
structural contract the traversal relies on, against the golden buffers dumped from the reference's own classes where the
two must coincide (quad triangulation, primitive and material layout), and by rendering them with the oracle."""
import os

import numpy as np
import pytest

from lens_trace_amd import scene as sc
from lens_trace_amd import synth
from oracle import pyoracle as po
from tests.conftest import GOLDEN

WALL_OBJ = """mtllib green_wall.mtl
o Plane
v -25.000000 -25.000000 -0.000001
v 25.000000 -25.000000 -0.000001
v -25.000000 25.000000 0.000001
v 25.000000 25.000000 0.000001
vt 0 0
vn 0.0000 -0.0000 1.0000
usemtl Material
s off
f 1/1/1 2/1/1 4/1/1 3/1/1
"""
WALL_MTL = """newmtl Material
Ns 323.999994
Ka 1.000000 1.000000 1.000000
Kd 0.000000 1.000000 0.000000
Ke 0.000000 0.000000 0.000000
Ni 1.450000
d 1.000000
illum 2
"""


def check_bvh_contract(s):
    """What the traversal assumes: pre-order layout (left child = i+1 < right child), every primitive in exactly one
    leaf, child boxes inside the parent's box, leaf boxes around their triangle."""
    nv, pv = s.node_view, s.prim_view
    n = s.n_nodes
    leaf = nv["primitiveCount"] > 0
    assert leaf.sum() == s.n_prims and (nv["primitiveCount"][leaf] == 1).all()
    assert n == 2 * s.n_prims - 1
    assert sorted(nv["offset"][leaf].tolist()) == list(range(s.n_prims))
    inner = np.flatnonzero(~leaf)
    right = nv["offset"][inner]
    assert (right > inner + 1).all() and (right < n).all()
    for child in (inner + 1, right):
        assert (nv["boundsMin"][child] >= nv["boundsMin"][inner]).all()
        assert (nv["boundsMax"][child] <= nv["boundsMax"][inner]).all()
    lo = np.minimum(np.minimum(pv["positionA"], pv["positionB"]), pv["positionC"])
    hi = np.maximum(np.maximum(pv["positionA"], pv["positionB"]), pv["positionC"])
    li = np.flatnonzero(leaf)
    assert np.array_equal(nv["boundsMin"][li], lo[nv["offset"][li]])
    assert np.array_equal(nv["boundsMax"][li], hi[nv["offset"][li]])
    # every node is reachable exactly once from the root
    seen = np.zeros(n, dtype=np.int32)
    stack = [0]
    while stack:
        i = stack.pop()
        seen[i] += 1
        if not leaf[i]:
            stack.append(int(nv["offset"][i]))
            stack.append(i + 1)
    assert (seen == 1).all()


def test_obj_loader_reproduces_the_reference_dump_of_green_wall(tmp_path):
    (tmp_path / "green_wall.obj").write_text(WALL_OBJ)
    (tmp_path / "green_wall.mtl").write_text(WALL_MTL)
    mine = sc.load_obj(tmp_path / "green_wall.obj").validate()
    ref = sc.load_ltsb(os.path.join(GOLDEN, "green_wall_O0.ltsb"))
    # same triangulation of the quad (ties on the 1-3 diagonal), same Primitive / Material / LightContainer bytes
    assert np.array_equal(mine.prims, ref.prims)
    assert np.array_equal(mine.materials, ref.materials)
    assert np.array_equal(mine.lights, ref.lights)
    # 3 nodes either way; the reference's leaf nodes carry uninitialised axis/pad bytes, so compare the fields
    for f in ("boundsMin", "boundsMax", "offset", "primitiveCount"):
        assert np.array_equal(mine.node_view[f], ref.node_view[f]), f
    check_bvh_contract(mine)
    img = po.render(mine, sc.camera_bytes(0, 2.5, -50), 100, 100, po.BASIC)
    assert np.array_equal(img.reshape(-1, 3), np.tile(np.float32([0, 1, 0]), (10000, 1)))


def test_obj_loader_rejects_what_the_renderer_cannot_use(tmp_path, capfd):
    (tmp_path / "m.mtl").write_text(WALL_MTL)
    (tmp_path / "no_normals.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl Material\nf 1 2 3\n")
    with pytest.raises(ValueError):
        sc.load_obj(tmp_path / "no_normals.obj")
    (tmp_path / "no_mtl.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    with pytest.raises(ValueError):
        sc.load_obj(tmp_path / "no_mtl.obj")
    with pytest.raises(ValueError):
        sc.load_obj(tmp_path / "does_not_exist.obj")
    capfd.readouterr()


def test_polygons_negative_indices_and_lights(tmp_path):
    mtl = WALL_MTL + "\nnewmtl Lamp\nKd 0.8 0.8 0.8\nKe 1 1 1\nNi 1.0\nd 1.0\n\nnewmtl Glass\nKd 1 1 1\nNi 1.5\nd 0.25\n"
    (tmp_path / "s.mtl").write_text(mtl)
    # an octagon (ear clipping -> 6 triangles), a triangle given with negative indices, an emissive quad
    pts = [(np.cos(a) * 3, np.sin(a) * 3, 0.0) for a in np.linspace(0, 2 * np.pi, 8, endpoint=False)]
    obj = "mtllib s.mtl\n" + "".join("v %f %f %f\n" % p for p in pts) + "vn 0 0 -1\nusemtl Material\nf " + " ".join("%d//1" % (i + 1) for i in range(8)) + "\n"
    obj += "v 5 0 1\nv 6 0 1\nv 5 1 1\nusemtl Glass\nf -3//-1 -2//-1 -1//-1\n"
    obj += "v -1 8 -1\nv 1 8 -1\nv 1 8 1\nv -1 8 1\nvn 0 -1 0\nusemtl Lamp\nf 12//2 13//2 14//2 15//2\n"
    (tmp_path / "s.obj").write_text(obj)
    s = sc.load_obj(tmp_path / "s.obj").validate()
    assert s.n_prims == 6 + 1 + 2
    check_bvh_contract(s)
    pv, mv = s.prim_view, s.material_view
    assert mv["dissolve"].tolist() == [1.0, 1.0, 0.25] and mv["ior"][2] == 1.5 and mv["emission"][1].tolist() == [1, 1, 1]
    lit = s.light_view[0]
    assert lit["count"] == 2 and all(pv["materialIndex"][p] == 1 for p in lit["primitives"][:2])
    # the octagon's triangles tile it: areas add up
    tri = pv[pv["materialIndex"] == 0]
    area = 0.5 * np.linalg.norm(np.cross(tri["positionB"] - tri["positionA"], tri["positionC"] - tri["positionA"]), axis=1).sum()
    assert abs(area - 2 * np.sqrt(2) * 9) < 1e-4


@pytest.mark.parametrize("make,tris", [(lambda: synth.heightfield_wall(64), 2 * 64 * 64 + 2), (lambda: synth.blob_in_box(3), None),
                                      (lambda: synth.triangle_soup(5000), 5004), (lambda: synth.colonnade(4, 24, 10), None)])
def test_builder_contract_on_synthetic_scenes(make, tris):
    s = make().validate()
    if tris is not None:
        assert s.n_prims == tris
    check_bvh_contract(s)
    assert s.height <= 64 and s.light_view[0]["count"] == 2
    assert s.light_view[0]["primitives"][0] != 0          # primitive 0 must not be emissive
    # deterministic: building twice gives identical bytes
    t = make()
    assert np.array_equal(s.nodes, t.nodes) and np.array_equal(s.prims, t.prims)


def test_builder_tree_is_balanced_and_renders():
    s = synth.heightfield_wall(96)
    assert s.height <= int(np.ceil(np.log2(s.n_prims))) + 1     # median split
    img, st = po.render(s, s.camera, 64, 36, po.ACCUMULATOR, threads=4, want_stats=True)
    assert st["max_stack"] <= s.height
    assert (img.sum(axis=2) > 0).mean() > 0.5


@pytest.mark.parametrize("make", [lambda b: synth.heightfield_wall(64, bvh=b), lambda b: synth.blob_in_box(3, bvh=b),
                                  lambda b: synth.triangle_soup(5000, bvh=b), lambda b: synth.colonnade(4, 24, 10, bvh=b)])
def test_sah_builder_emits_the_same_layout_with_fewer_node_visits(make):
    """ACCELERATION_STRUCTURE_TYPE_BVH_SAH (this backend's addition): same buffers contract, one triangle per
    leaf, height bounded by ceil(log2 n) + 4, same triangles -- and the reference's traversal visits fewer nodes in it."""
    med, sah = make(sc.BVH_MEDIAN).validate(), make(sc.BVH_SAH).validate()
    check_bvh_contract(sah)
    assert sah.n_prims == med.n_prims and sah.n_nodes == med.n_nodes
    assert sah.height <= int(np.ceil(np.log2(sah.n_prims))) + 4 and sah.height <= 64
    assert np.array_equal(sah.materials, med.materials) and sah.light_view[0]["count"] == med.light_view[0]["count"]
    # the same set of triangles, in another order
    key = lambda s: np.sort(np.ascontiguousarray(s.prims).reshape(-1, 76).copy().view("V76").reshape(-1))   # noqa: E731
    assert np.array_equal(key(sah), key(med))
    again = make(sc.BVH_SAH)
    assert np.array_equal(sah.nodes, again.nodes) and np.array_equal(sah.prims, again.prims)     # deterministic
    _, sm = po.render(med, med.camera, 64, 36, po.ACCUMULATOR, threads=4, want_stats=True)
    img, ss = po.render(sah, sah.camera, 64, 36, po.ACCUMULATOR, threads=4, want_stats=True)
    assert ss["max_stack"] <= sah.height
    assert ss["node_visits"] < sm["node_visits"]
    assert (img.sum(axis=2) > 0).mean() > 0.3
