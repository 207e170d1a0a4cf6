"""CPU checks (no GPU) of what lt_hip_set_scene derives from a caller's BVH for its walks (lens_trace_amd/csrc/lt_retree.hpp,
lt_walk_asm.hpp), through the host-only entry point lt_hip_own_hierarchy:

* the backend's own hierarchy keeps every leaf of the caller's tree bit for bit (box, primitive offset), nests, is in the
  caller's pre-order layout and respects its height bound; a tree whose boxes do not nest gets none;
* the order table (rank8) is the reference's depth-first, near-child-first leaf order (acc.cl:150-160) for each of the eight
  direction-sign octants;
* the packet walks' conservative interior test accepts whenever the reference's slab test (acc.cl:113-130) does, on random and
  on adversarial (boundary-grazing) rays -- the inequality proved in lt_walk_asm.hpp, tried in float32 arithmetic."""
import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd import synth


def view(s):
    return s.node_view


def leaves_of(nodes):
    lv = nodes[nodes["primitiveCount"] != 0]
    key = np.concatenate([lv["boundsMin"].view(np.uint32), lv["boundsMax"].view(np.uint32), lv["offset"].astype(np.uint32)[:, None]], axis=1)
    return key[np.lexsort(key.T[::-1])]


def check_tree(own, height):
    n = len(own)
    depth = np.zeros(n, dtype=np.int64)
    seen = np.zeros(n, dtype=bool)
    seen[0] = True
    for i in range(n):                                   # pre-order: children sit behind their parent
        assert seen[i], "unreachable node in the own hierarchy"
        if own["primitiveCount"][i] == 0:
            l, r = i + 1, int(own["offset"][i])
            assert i + 1 < r < n and own["axis"][i] <= 2
            for c in (l, r):
                assert not seen[c]
                seen[c] = True
                depth[c] = depth[i] + 1
                assert np.all(own["boundsMin"][c] >= own["boundsMin"][i]) and np.all(own["boundsMax"][c] <= own["boundsMax"][i])
            # an interior box is the union of its children's
            assert np.array_equal(np.minimum(own["boundsMin"][l], own["boundsMin"][r]), own["boundsMin"][i])
            assert np.array_equal(np.maximum(own["boundsMax"][l], own["boundsMax"][r]), own["boundsMax"][i])
    assert int(depth.max()) == height


@pytest.mark.parametrize("name,slack", [("cornell", 2), ("wall", 0), ("wall", 2), ("soup", 2), ("blob", 3), ("wall_sah", 2), ("wall", -1)])
def test_own_hierarchy_keeps_the_leaves_and_nests(name, slack):
    import os
    from tests.conftest import GOLDEN
    s = {"cornell": lambda: sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")), "wall": lambda: synth.heightfield_wall(40),
         "soup": lambda: synth.triangle_soup(5000), "blob": lambda: synth.blob_in_box(3),
         "wall_sah": lambda: synth.heightfield_wall(40, bvh=sc.BVH_SAH)}[name]().validate()
    nodes = view(s)
    h, own, _ = C.own_hierarchy(nodes, s.n_prims, slack)
    n_leaves = int((nodes["primitiveCount"] != 0).sum())
    assert h >= 0 and len(own) == 2 * n_leaves - 1
    assert np.array_equal(leaves_of(own), leaves_of(nodes))            # every leaf, bit for bit: box and primitive offset
    check_tree(own, h)
    if slack >= 0:
        assert h <= int(np.ceil(np.log2(n_leaves))) + slack
        h2, own2, _ = C.own_hierarchy(nodes, s.n_prims, slack)
        assert h2 == h and own2.tobytes() == own.tobytes()             # deterministic
    else:                                                               # the caller's own splits: the same tree
        assert own.tobytes() == nodes.tobytes()


def test_the_tree_does_not_depend_on_the_thread_count(monkeypatch):
    s = synth.heightfield_wall(180).validate()            # 64 800 leaves: above the size from which the build goes parallel
    nodes = view(s)
    trees = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("LT_RETREE_THREADS", threads)
        h, own, _ = C.own_hierarchy(nodes, s.n_prims, 2)
        trees.append((h, own.tobytes()))
    assert trees[0] == trees[1] == trees[2]


def test_surface_area_sum_drops():
    """Without clipping against the closest hit the expected number of nodes a random ray visits is the sum of the nodes' surface
    areas over the root's: what the build minimises."""
    def cost(t):
        d = (t["boundsMax"] - t["boundsMin"]).astype(np.float64)
        a = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
        return a.sum() / a[0]
    for s in (synth.heightfield_wall(64).validate(), synth.blob_in_box(3).validate()):
        nodes = view(s)
        _, own, _ = C.own_hierarchy(nodes, s.n_prims, 2)
        assert cost(own) < 0.8 * cost(nodes)


def test_a_tree_whose_boxes_do_not_nest_gets_none():
    s = synth.heightfield_wall(8).validate()
    nodes = view(s).copy()
    assert C.own_hierarchy(nodes, s.n_prims, 2)[0] >= 0
    child = int(np.flatnonzero(nodes["primitiveCount"] != 0)[3])
    broken = nodes.copy()
    broken["boundsMax"][child, 1] += 100.0                               # a leaf that pokes out of its ancestors
    assert C.own_hierarchy(broken, s.n_prims, 2)[0] == -1
    nan = nodes.copy()
    nan["boundsMin"][child, 0] = np.nan
    assert C.own_hierarchy(nan, s.n_prims, 2)[0] == -1
    huge = nodes.copy()
    huge["boundsMax"][0, 2] = 2.0 ** 41                                  # beyond the magnitude the conservative test is proved for
    assert C.own_hierarchy(huge, s.n_prims, 2)[0] == -1
    single = nodes[nodes["primitiveCount"] != 0][:1].copy()              # a one-leaf tree: nothing to build
    assert C.own_hierarchy(single, s.n_prims, 2)[0] == -1


def test_rank8_is_the_references_leaf_order():
    s = synth.blob_in_box(2).validate()
    nodes = view(s)
    _, _, ranks = C.own_hierarchy(nodes, s.n_prims, 2, want_ranks=True)
    for octant in range(8):
        order = []
        stack = [0]
        while stack:                                                     # acc.cl:132-171: near child first, by dirIsNeg[node->axis]
            i = stack.pop()
            if nodes["primitiveCount"][i] != 0:
                order.append(int(nodes["offset"][i]))
                continue
            neg = (octant >> int(nodes["axis"][i])) & 1
            near, far = (int(nodes["offset"][i]), i + 1) if neg else (i + 1, int(nodes["offset"][i]))
            stack.append(far)
            stack.append(near)
        want = np.full(s.n_prims, 0xFFFFFFFF, dtype=np.uint32)
        for k, p in enumerate(order):
            if want[p] == 0xFFFFFFFF:
                want[p] = k
        assert np.array_equal(ranks[:, octant], want)


# ---- the conservative interior test of the packet walks ------------------------------------------------------------------
f32 = np.float32


def fma32(a, b, c):
    """fl32(a * b + c) with one rounding: a * b is exact in float64 (24 + 24 bits); the sum's double rounding does not matter at
    the margins tested here."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def outwards(b, up):
    """lt_outwards (lt_capi.hip): 2^-21 of the bound, then one float further."""
    t = (b + np.abs(b) * f32(2.0 ** -21)) if up else (b - np.abs(b) * f32(2.0 ** -21))
    t = t.astype(f32)
    return np.nextafter(t, f32(np.inf) if up else f32(-np.inf)).astype(f32)


def reference_test(lo, hi, o, inv):
    t0 = ((lo - o).astype(f32) * inv).astype(f32)
    t1 = ((hi - o).astype(f32) * inv).astype(f32)
    t_enter = np.minimum(t0, t1).max(axis=-1)
    t_exit = np.maximum(t0, t1).min(axis=-1)
    return (t_enter <= t_exit) & (t_exit > 0)


def conservative_test(lo, hi, o, inv):
    p = (o * inv).astype(f32)
    mg = (np.abs(p).max(axis=-1) * f32(2.0 ** -19) + f32(2.0 ** -140)).astype(f32)
    t0 = fma32(outwards(lo, False), inv, -p)
    t1 = fma32(outwards(hi, True), inv, -p)
    t_enter = np.minimum(t0, t1).max(axis=-1)
    t_exit = np.maximum(t0, t1).min(axis=-1)
    return (t_exit + mg).astype(f32) >= np.maximum(t_enter, np.float32(1e-45))


@pytest.mark.parametrize("scale", [1e-3, 1.0, 50.0, 1e6])
def test_conservative_test_accepts_whenever_the_reference_does(scale):
    rng = np.random.default_rng(7)
    n = 400000
    c = rng.uniform(-scale, scale, (n, 3))
    half = np.abs(rng.normal(0, scale * 0.05, (n, 3))) * rng.choice([0.0, 1e-4, 1.0], (n, 3))     # flat and sliver boxes too
    lo, hi = (c - half).astype(f32), (c + half).astype(f32)
    o = rng.uniform(-scale, scale, (n, 3)).astype(f32) * rng.choice([1.0, 10.0, 1000.0], (n, 1)).astype(f32)
    # aim at a point ON the box's surface or edges (a grazing ray), then perturb by a few ulps
    w = rng.choice([0.0, 1.0, 0.5], (n, 3))
    target = (lo * (1 - w) + hi * w).astype(f32)
    d = (target - o).astype(f32)
    d = (d * (1 + rng.integers(-3, 4, (n, 3)) * 2.0 ** -23)).astype(f32)
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        inv = (f32(1.0) / d).astype(f32)
        ok = np.isfinite(inv).all(axis=-1) & (np.abs(inv) < 2.0 ** 60).all(axis=-1)
        ref = reference_test(lo, hi, o, inv)
        con = conservative_test(lo, hi, o, inv)
    assert ok.sum() > n // 2 and ref[ok].sum() > n // 20
    missed = ok & ref & ~con
    assert not missed.any(), "%d rays pass the reference's slab test and fail the conservative one" % int(missed.sum())
    # ... and it is not vacuous: on rays that do not graze, what it lets through beyond the reference is next to nothing
    d = rng.normal(0, 1, (n, 3)).astype(f32)
    inv = (f32(1.0) / d).astype(f32)
    ok = (np.abs(inv) < 2.0 ** 60).all(axis=-1)
    ref, con = reference_test(lo, hi, o, inv), conservative_test(lo, hi, o, inv)
    assert not (ok & ref & ~con).any()
    assert (ok & con & ~ref).sum() <= 1e-3 * ok.sum() + 5
