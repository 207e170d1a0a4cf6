"""CPU checks (no GPU) of what lt_hip_set_scene derives from a caller's BVH for its walks (lens_trace_amd/csrc/lt_retree.hpp,
lt_walk_asm.hpp), through the host-only entry point lt_hip_own_hierarchy:

* the backend's own hierarchy keeps every leaf of the caller's tree bit for bit (box, primitive offset), nests, is in the
  caller's pre-order layout and respects its height bound; a tree whose boxes do not nest gets none;
* the order table (rank8) is the reference's depth-first, near-child-first leaf order (acc.cl:150-160) for each of the eight
  direction-sign octants;
* the packet walks' conservative interior test accepts whenever the reference's slab test (acc.cl:113-130) does, on random and
  on adversarial (boundary-grazing) rays -- the inequality proved in lt_walk_asm.hpp, tried in float32 arithmetic;
* the per-lane walks' 4-wide groups (lt_hip_own_wide) hold every node of the tree once, their 16-bit child boxes enclose the
  true ones, and the test on a quantised box accepts whenever the reference's accepts the true box (lt_device.hpp)."""
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


# ---- the per-lane walks' 16-byte quantised nodes (lt_own16.hpp, traverse_own_lane) --------------------------------------------
def chain_tree(lo, hi):
    """A caller's tree over n leaf boxes in the reference's layout: a right-leaning chain (interior node 2k: left child = leaf
    2k + 1, right child = the rest), every interior box the union of what is below it.  Leaf k refers to primitive k."""
    n = len(lo)
    nodes = np.zeros(2 * n - 1, dtype=sc.NODE_DTYPE)
    sufmin = np.minimum.accumulate(lo[::-1], axis=0)[::-1]
    sufmax = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
    for k in range(n - 1):
        i = 2 * k
        nodes["boundsMin"][i], nodes["boundsMax"][i] = sufmin[k], sufmax[k]
        nodes["offset"][i] = i + 2
        nodes["boundsMin"][i + 1], nodes["boundsMax"][i + 1] = lo[k], hi[k]
        nodes["offset"][i + 1], nodes["primitiveCount"][i + 1] = k, 1
    nodes["boundsMin"][-1], nodes["boundsMax"][-1] = lo[-1], hi[-1]
    nodes["offset"][-1], nodes["primitiveCount"][-1] = n - 1, 1
    return nodes


def random_boxes(rng, n, scale, centre):
    c = centre + rng.uniform(-scale, scale, (n, 3))
    half = np.abs(rng.normal(0, scale * 0.02, (n, 3))) * rng.choice([0.0, 1e-4, 1.0], (n, 3))     # flat and sliver boxes too
    return (c - half).astype(f32), (c + half).astype(f32)


def wide_children(own, slots, n_prims):
    """The binary node behind every child slot of every group, by walking the groups from the root: (child[g, k] or -1,
    group_of_node) -- and the structural checks on the way."""
    G = len(slots)
    leaf = own["primitiveCount"] != 0
    child = np.full((G, 4), -1, dtype=np.int64)
    node_of_group = np.full(G, -1, dtype=np.int64)
    node_of_group[0] = 0
    leaf_node_of_prim = {int(own["offset"][i]): i for i in np.flatnonzero(leaf)}
    seen_leaves = 0
    for g in range(G):                                   # depth-first numbering: a group's children have larger numbers
        b = int(node_of_group[g])
        assert b >= 0 and not leaf[b], "group %d is not reachable from the root" % g
        # the (up to four) nodes the group must hold: b's two children, with up to two of them replaced by their own children
        for k in range(4):
            link = int(slots["link"][g, k])
            if link == (0x80000000 | (G + n_prims)):      # empty slot: a box no ray enters
                assert tuple(slots["q"][g, k]) == (65535, 65535, 65535, 0, 0, 0)
                continue
            if link & 0x80000000:
                prim = (link & 0x7fffffff) - G
                assert 0 <= prim < n_prims
                child[g, k] = leaf_node_of_prim[prim]
                seen_leaves += 1
            else:
                assert g < link < G and node_of_group[link] < 0
                child[g, k] = -2                          # resolved below
        # which interior nodes do the linked groups stand for?  b's descendants within two dissolves, in slot order
        frontier = [b + 1, int(own["offset"][b])]
        want = set()
        for _ in range(2):
            cand = [x for x in frontier if not leaf[x]]
            if not cand or len(frontier) >= 4:
                break
            ext = (own["boundsMax"][cand].astype(np.float32) - own["boundsMin"][cand].astype(np.float32))
            area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
            x = cand[int(np.argmax(area))]
            frontier[frontier.index(x)] = x + 1
            frontier.append(int(own["offset"][x]))
        interior = [x for x in frontier if not leaf[x]]
        leaves = [x for x in frontier if leaf[x]]
        ks = [k for k in range(4) if child[g, k] == -2]
        assert len(ks) == len(interior) and sorted(int(c) for c in child[g] if c >= 0) == sorted(leaves)
        for k, x in zip(ks, interior):
            child[g, k] = x
            node_of_group[int(slots["link"][g, k])] = x
        # leaves sit behind the interior children
        kinds = [bool(leaf[c]) for c in child[g] if c >= 0]
        assert kinds == sorted(kinds)
    assert seen_leaves == int(leaf.sum())
    return child


@pytest.mark.parametrize("scale,centre", [(1.0, 0.0), (1e-3, 0.0), (50.0, 10.0), (1.0, 3000.0), (1e6, 0.0)])
def test_wide_groups_hold_the_tree_and_their_slots_enclose_the_boxes(scale, centre):
    rng = np.random.default_rng(11)
    lo, hi = random_boxes(rng, 3000, scale, centre)
    h, own, _ = C.own_hierarchy(chain_tree(lo, hi), len(lo), 2)
    assert h >= 0
    hw, O, S, slots = C.own_wide(own, len(lo))
    child = wide_children(own, slots, len(lo))
    assert 0 < hw <= h and len(slots) <= len(lo) - 1
    used = child >= 0
    q = slots["q"][used].astype(np.float64)
    L = O.astype(np.float64) + q[:, :3] * S.astype(np.float64)
    H = O.astype(np.float64) + q[:, 3:] * S.astype(np.float64)
    blo, bhi = own["boundsMin"][child[used]].astype(np.float64), own["boundsMax"][child[used]].astype(np.float64)
    u = 2.0 ** -24
    assert np.all(L <= blo - 8 * u * np.abs(blo)) and np.all(H >= bhi + 8 * u * np.abs(bhi))
    # ... and tightly: at most two grid steps and 2^-20 of the bound away
    assert np.all(blo - L <= 2 * S + 2.0 ** -20 * np.abs(blo) + 1e-44) and np.all(H - bhi <= 2 * S + 2.0 ** -20 * np.abs(bhi) + 1e-44)


def test_two_leaves_on_one_primitive_get_no_wide_groups():
    rng = np.random.default_rng(2)
    lo, hi = random_boxes(rng, 50, 1.0, 0.0)
    nodes = chain_tree(lo, hi)
    leaves = np.flatnonzero(nodes["primitiveCount"] != 0)
    nodes["offset"][leaves[7]] = nodes["offset"][leaves[3]]
    h, own, _ = C.own_hierarchy(nodes, len(lo), 2)
    assert h >= 0
    with pytest.raises(C.LensTraceError):
        C.own_wide(own, len(lo))


def own16_test(O, S, q, o, inv):
    """own16_ray + own16_box_test (lt_device.hpp) in float32 arithmetic: per-axis margins folded into the two constants, the
    near / far bound picked by the direction's sign."""
    p = (o * inv).astype(f32)
    sI = (S[None, :] * inv).astype(f32)
    Ob = np.broadcast_to(O, inv.shape).astype(f32)
    c = fma32(Ob, inv, -p)
    k = (np.abs(((f32(65535.0) * S[None, :]).astype(f32) * inv).astype(f32)) + np.abs((Ob * inv).astype(f32))).astype(f32)
    k = (k + np.abs(p)).astype(f32)
    m = ((k * f32(2.0 ** -21)).astype(f32) + f32(2.0 ** -140)).astype(f32)
    cN, cF = (c - m).astype(f32), (c + m).astype(f32)
    neg = inv < 0
    ql, qh = q[:, :3].astype(f32), q[:, 3:].astype(f32)
    t_enter = fma32(np.where(neg, qh, ql), sI, cN).max(axis=-1)
    t_exit = fma32(np.where(neg, ql, qh), sI, cF).min(axis=-1)
    return t_exit >= np.maximum(t_enter, np.float32(1e-45))


@pytest.mark.parametrize("scale,centre", [(1.0, 0.0), (1e-3, 0.0), (50.0, 10.0), (1.0, 3000.0), (1e6, 0.0)])
def test_quantised_test_accepts_whenever_the_reference_accepts_the_true_box(scale, centre):
    rng = np.random.default_rng(5)
    lo, hi = random_boxes(rng, 4000, scale, centre)
    _, own, _ = C.own_hierarchy(chain_tree(lo, hi), len(lo), 2)
    _, O, S, slots = C.own_wide(own, len(lo))
    child = wide_children(own, slots, len(lo))
    used = child >= 0
    reps = 60
    idx = np.tile(np.arange(int(used.sum())), reps)
    n = len(idx)
    blo, bhi, q = own["boundsMin"][child[used]][idx], own["boundsMax"][child[used]][idx], slots["q"][used][idx]
    o = (centre + rng.uniform(-scale, scale, (n, 3)) * rng.choice([1.0, 3.0, 100.0], (n, 1))).astype(f32)
    w = rng.choice([0.0, 1.0, 0.5], (n, 3))                       # aim at the box's faces, edges and corners, a few ulps off
    target = (blo * (1 - w) + bhi * w).astype(f32)
    d = (target - o).astype(f32)
    d = (d * (1 + rng.integers(-3, 4, (n, 3)) * 2.0 ** -23)).astype(f32)
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        inv = (f32(1.0) / d).astype(f32)
        ok = np.isfinite(inv).all(axis=-1) & (np.abs(inv) < 2.0 ** 60).all(axis=-1)
        ref = reference_test(blo, bhi, o, inv)
        con = own16_test(O, S, q, o, inv)
    assert ok.sum() > n // 2 and ref[ok].sum() > n // 20
    missed = ok & ref & ~con
    assert not missed.any(), "%d rays pass the reference's slab test of a box and fail the test of its quantised node" % int(missed.sum())
    # not vacuous: random rays that miss the true box by more than a few grid steps miss the quantised node too
    d = rng.normal(0, 1, (n, 3)).astype(f32)
    inv = (f32(1.0) / d).astype(f32)
    ok = (np.abs(inv) < 2.0 ** 60).all(axis=-1)
    ref, con = reference_test(blo, bhi, o, inv), own16_test(O, S, q, o, inv)
    assert not (ok & ref & ~con).any()
    # (a scene far from the origin relative to its size pays for it: the boxes' outward push and the test's margin are relative
    # to the coordinates' magnitude, 2^-21 of 3000 against boxes of 0.02 -- looser tests, never wrong ones)
    assert (ok & con & ~ref).sum() <= (1.0 if centre > 100.0 else 0.02) * max(1, (ok & ref).sum()) + 5


def test_box_unions_take_one_order_of_the_floats_and_the_median_rule_is_a_function_of_the_set():
    """What makes the device-side build (lt_prep.hip, held against this one byte for byte on the GPU) possible: (1) unions are taken
    in ONE total order of the floats, -0 below +0 (the integers the kernels' atomic min / max work on), so an interior bound's zero
    has the sign that order gives it whatever the order of the leaves; (2) where no plane separates the centroids (coincident
    ones; two leaves) the count / 2 smallest (centroid, leaf index) go left and both halves keep the order they stood in -- the
    same tree for any permutation of equal-centroid leaves' boxes that keeps their indices."""
    # (1) signed zeros: leaves whose bounds are +0.0 and -0.0 on one axis, in both orders
    lo = np.array([[-0.0, 1.0, 1.0], [0.0, 2.0, 2.0], [0.5, 3.0, 3.0], [-0.0, 4.0, 4.0]], dtype=f32)
    hi = np.array([[0.0, 1.5, 1.5], [-0.0, 2.5, 2.5], [0.75, 3.5, 3.5], [0.0, 4.5, 4.5]], dtype=f32)
    roots = []
    for perm in ([0, 1, 2, 3], [1, 0, 3, 2], [3, 2, 1, 0]):
        h, own, _ = C.own_hierarchy(chain_tree(lo[perm], hi[perm]), 4, 2)
        assert h >= 1
        roots.append((own["boundsMin"][0].view(np.uint32).tolist(), own["boundsMax"][0].view(np.uint32).tolist()))
    assert roots[0] == roots[1] == roots[2]
    assert roots[0][0][0] == 0x80000000 and roots[0][1][0] == 0x3f400000      # min = -0.0 (below +0.0), max = 0.75
    # (2) coincident centroids: 64 boxes around one point (different extents), then around two points
    rng = np.random.default_rng(4)
    for centres in (np.zeros((64, 3)), np.repeat(np.array([[0.0, 0, 0], [3.0, 1, 2]]), 32, axis=0)):
        half = rng.uniform(0.1, 1.0, (64, 3))
        blo, bhi = (centres - half).astype(f32), (centres + half).astype(f32)
        assert np.array_equal((0.5 * blo + 0.5 * bhi).astype(f32), centres.astype(f32))     # exactly coincident centroids
        h, own, _ = C.own_hierarchy(chain_tree(blo, bhi), 64, 2)
        assert 6 <= h <= 8
        check_tree(own, h)
        # leaves under the root's left child: the first half of the caller's order (coincident: "the first half as it stands")
        leaf = own["primitiveCount"] != 0
        if np.all(centres == 0):
            left = np.arange(1, int(own["offset"][0]))
            assert sorted(own["offset"][left][leaf[left]].tolist()) == list(range(32))
        h2, own2, _ = C.own_hierarchy(chain_tree(blo, bhi), 64, 2)
        assert np.array_equal(own.view(np.uint8), own2.view(np.uint8))
