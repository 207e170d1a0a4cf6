"""Synthetic scenes for the BASELINE.json configurations (there are no bunny / Sponza assets in the reference and
no network): closed-form or fixed-seed geometry, smooth vertex normals, one emissive quad (2 triangles), every
triangle with a material -- the preconditions of the reference's loader.  All scenes are framed for
the reference camera (0, 2.5, -50), whose view at z = 0 spans x in [-4.5, 4.5], y in [-2, 7]."""
import numpy as np

from .scene import MATERIAL_DTYPE, build_from_triangles, camera_bytes


def _materials(colors):
    m = np.zeros(len(colors) + 1, dtype=MATERIAL_DTYPE)
    m["ior"] = 1.45
    m["dissolve"] = 1.0
    for i, c in enumerate(colors):
        m[i]["diffuse"] = c
    m[-1]["diffuse"] = (0.8, 0.8, 0.8)
    m[-1]["emission"] = (1.0, 1.0, 1.0)     # last material = the light
    return m


def _light_quad(x0, x1, y, z0, z1):
    """Two emissive triangles facing -y, above and in front of the geometry, outside the camera's view."""
    p = np.float32([[[x0, y, z0], [x1, y, z0], [x1, y, z1]], [[x0, y, z0], [x1, y, z1], [x0, y, z1]]])
    n = np.tile(np.float32([0, -1, 0]), (2, 3, 1))
    return p, n


def _grid_triangles(P, N):
    """P, N: [ny+1, nx+1, 3] vertex positions / normals -> two triangles per cell."""
    a, b, c, d = P[:-1, :-1], P[:-1, 1:], P[1:, 1:], P[1:, :-1]
    na, nb, nc, nd = N[:-1, :-1], N[:-1, 1:], N[1:, 1:], N[1:, :-1]
    pos = np.stack([np.stack([a, b, c], axis=2), np.stack([a, c, d], axis=2)], axis=2).reshape(-1, 3, 3)
    nrm = np.stack([np.stack([na, nb, nc], axis=2), np.stack([na, nc, nd], axis=2)], axis=2).reshape(-1, 3, 3)
    return pos.astype(np.float32), nrm.astype(np.float32)


def heightfield_wall(cells=708, camera=None, bvh=None):
    """Config 4: a cells x cells height-field wall over x in [-5,5], y in [-2.5,7.5] that fills the frame
    (2*cells^2 triangles: 708 -> 1 002 528) plus a 2-triangle light.  Closed-form sines, no RNG."""
    x = np.linspace(-5.0, 5.0, cells + 1)
    y = np.linspace(-2.5, 7.5, cells + 1)
    X, Y = np.meshgrid(x, y)
    Z = 0.35 * np.sin(1.7 * X) * np.sin(1.3 * Y) + 0.08 * np.sin(9.0 * X + 5.0 * Y) + 0.03 * np.sin(23.0 * X - 17.0 * Y)
    dZdx = 0.35 * 1.7 * np.cos(1.7 * X) * np.sin(1.3 * Y) + 0.08 * 9.0 * np.cos(9.0 * X + 5.0 * Y) + 0.03 * 23.0 * np.cos(23.0 * X - 17.0 * Y)
    dZdy = 0.35 * 1.3 * np.sin(1.7 * X) * np.cos(1.3 * Y) + 0.08 * 5.0 * np.cos(9.0 * X + 5.0 * Y) - 0.03 * 17.0 * np.cos(23.0 * X - 17.0 * Y)
    P = np.stack([X, Y, Z], axis=-1)
    N = np.stack([dZdx, dZdy, -np.ones_like(Z)], axis=-1)     # facing the camera (-z)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    pos, nrm = _grid_triangles(P, N)
    # three colour bands so the image is not monochrome
    band = (np.arange(pos.shape[0]) // (2 * cells)) * 3 // cells
    lp, ln = _light_quad(-1.5, 1.5, 9.5, -9.0, -6.0)
    mats = _materials([(0.8, 0.8, 0.8), (0.9, 0.3, 0.25), (0.3, 0.75, 0.35)])
    pos = np.concatenate([pos, lp])
    nrm = np.concatenate([nrm, ln])
    mi = np.concatenate([band.astype(np.int32), np.full(2, len(mats) - 1, dtype=np.int32)])
    return build_from_triangles(pos, nrm, mi, mats, camera or camera_bytes(0.0, 2.5, -50.0), bvh)


def triangle_soup(count=1000000, seed=1, camera=None, bvh=None):
    """Config 4 variant with incoherent traversal: `count` small random triangles in a slab in front of a back
    wall, fixed seed."""
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(-5, 5, count), rng.uniform(-2.5, 7.5, count), rng.uniform(-3.0, 0.0, count)], axis=-1)
    e = rng.normal(0.0, 0.03, (count, 3, 3))
    pos = (c[:, None, :] + e).astype(np.float32)
    n = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-20)
    n *= np.where(n[:, 2:3] > 0, -1.0, 1.0)                  # face the camera
    nrm = np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)
    wall_p = np.float32([[[-6, -3.5, 0.2], [6, -3.5, 0.2], [6, 8.5, 0.2]], [[-6, -3.5, 0.2], [6, 8.5, 0.2], [-6, 8.5, 0.2]]])
    wall_n = np.tile(np.float32([0, 0, -1]), (2, 3, 1))
    lp, ln = _light_quad(-1.5, 1.5, 9.5, -9.0, -6.0)
    mats = _materials([(0.8, 0.8, 0.8), (0.9, 0.3, 0.25), (0.3, 0.75, 0.35), (0.3, 0.4, 0.9)])
    mi = np.concatenate([rng.integers(0, 4, count).astype(np.int32), np.zeros(2, np.int32), np.full(2, len(mats) - 1, np.int32)])
    return build_from_triangles(np.concatenate([pos, wall_p, lp]), np.concatenate([nrm, wall_n, ln]), mi, mats,
                                camera or camera_bytes(0.0, 2.5, -50.0), bvh)


def wall_and_soup(cells=500, count=500000, seed=1, camera=None, bvh=None):
    """A mixed scene: the height-field wall over the whole frame with a triangle soup floating in front of its left half --
    coherent and incoherent shadow rays in one image (the per-wavefront choice of the shadow-ray walk)."""
    x = np.linspace(-5.0, 5.0, cells + 1)
    y = np.linspace(-2.5, 7.5, cells + 1)
    X, Y = np.meshgrid(x, y)
    Z = 0.35 * np.sin(1.7 * X) * np.sin(1.3 * Y) + 0.08 * np.sin(9.0 * X + 5.0 * Y)
    dZdx = 0.35 * 1.7 * np.cos(1.7 * X) * np.sin(1.3 * Y) + 0.08 * 9.0 * np.cos(9.0 * X + 5.0 * Y)
    dZdy = 0.35 * 1.3 * np.sin(1.7 * X) * np.cos(1.3 * Y) + 0.08 * 5.0 * np.cos(9.0 * X + 5.0 * Y)
    P = np.stack([X, Y, Z], axis=-1)
    N = np.stack([dZdx, dZdy, -np.ones_like(Z)], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    wp, wn = _grid_triangles(P, N)
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(-5, 0, count), rng.uniform(-2.5, 7.5, count), rng.uniform(-3.5, -0.8, count)], axis=-1)
    e = rng.normal(0.0, 0.03, (count, 3, 3))
    sp = (c[:, None, :] + e).astype(np.float32)
    n = np.cross(sp[:, 1] - sp[:, 0], sp[:, 2] - sp[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-20)
    n *= np.where(n[:, 2:3] > 0, -1.0, 1.0)
    sn = np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)
    lp, ln = _light_quad(-1.5, 1.5, 9.5, -9.0, -6.0)
    mats = _materials([(0.8, 0.8, 0.8), (0.9, 0.3, 0.25), (0.3, 0.75, 0.35)])
    mi = np.concatenate([np.zeros(len(wp), np.int32), rng.integers(0, 3, count).astype(np.int32), np.full(2, len(mats) - 1, np.int32)])
    return build_from_triangles(np.concatenate([wp, sp, lp]), np.concatenate([wn, sn, ln]), mi, mats,
                                camera or camera_bytes(0.0, 2.5, -50.0), bvh)


def blob_in_box(subdiv=5, camera=None, bvh=None):
    """Config 3: a displaced, subdivided sphere (20 * 4^subdiv... here a lat-long sphere of ~70 k triangles at the
    default) inside a 5-wall box with a 2-triangle light."""
    nu = nv = int(round(np.sqrt(70000 / 2)))                  # ~187 x 187 cells -> ~70 k triangles
    if subdiv != 5:
        nu = nv = max(8, int(round(np.sqrt(70000 / 2) * 2.0 ** (subdiv - 5))))
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1)
    v = np.linspace(1e-3, np.pi - 1e-3, nv + 1)
    U, V = np.meshgrid(u, v)
    R = 2.3 + 0.25 * np.sin(5 * U) * np.sin(4 * V) + 0.09 * np.sin(13 * U + 3.0) * np.sin(11 * V)
    P = np.stack([R * np.sin(V) * np.cos(U), 2.5 + R * np.cos(V), -2.5 + R * np.sin(V) * np.sin(U)], axis=-1)
    dU = np.gradient(P, axis=1)
    dV = np.gradient(P, axis=0)
    N = np.cross(dV, dU)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-20)
    pos, nrm = _grid_triangles(P, N)
    quads = []

    def quad(a, b, c, d, n):
        quads.append((np.float32([[a, b, c], [a, c, d]]), np.tile(np.float32(n), (2, 3, 1))))

    quad([-4.2, -1.8, -6], [4.2, -1.8, -6], [4.2, -1.8, 1], [-4.2, -1.8, 1], [0, 1, 0])        # floor
    quad([-4.2, 6.8, -6], [4.2, 6.8, -6], [4.2, 6.8, 1], [-4.2, 6.8, 1], [0, -1, 0])           # ceiling
    quad([-4.2, -1.8, 1], [4.2, -1.8, 1], [4.2, 6.8, 1], [-4.2, 6.8, 1], [0, 0, -1])           # back
    quad([-4.2, -1.8, -6], [-4.2, -1.8, 1], [-4.2, 6.8, 1], [-4.2, 6.8, -6], [1, 0, 0])        # left
    quad([4.2, -1.8, -6], [4.2, -1.8, 1], [4.2, 6.8, 1], [4.2, 6.8, -6], [-1, 0, 0])           # right
    wp = np.concatenate([q[0] for q in quads])
    wn = np.concatenate([q[1] for q in quads])
    lp, ln = _light_quad(-1.0, 1.0, 6.7, -5.5, -4.0)
    mats = _materials([(0.8, 0.8, 0.8), (0.9, 0.3, 0.25), (0.3, 0.75, 0.35), (0.85, 0.75, 0.3)])
    mi = np.concatenate([np.full(pos.shape[0], 3, np.int32), np.int32([0, 0, 0, 0, 0, 0, 1, 1, 2, 2]),
                         np.full(2, len(mats) - 1, np.int32)])
    return build_from_triangles(np.concatenate([pos, wp, lp]), np.concatenate([nrm, wn, ln]), mi, mats,
                                camera or camera_bytes(0.0, 2.5, -50.0), bvh)


def colonnade(columns=14, segments=96, rings=40, camera=None, bvh=None):
    """Config 5: a Sponza-like hall (~250 k triangles at the defaults): two rows of fluted columns carrying arches,
    floor, back wall and ceiling strips, one 2-triangle light.  Closed form, no RNG."""
    parts_p, parts_n, parts_m = [], [], []

    def add_grid(P, N, mat):
        p, n = _grid_triangles(P, N)
        parts_p.append(p)
        parts_n.append(n)
        parts_m.append(np.full(p.shape[0], mat, np.int32))

    # fluted columns: radius modulated by cos(16 theta), axis along y
    th = np.linspace(0.0, 2.0 * np.pi, segments + 1)
    yy = np.linspace(-2.0, 4.2, rings + 1)
    TH, YY = np.meshgrid(th, yy)
    for side in (-1.0, 1.0):
        for k in range(columns // 2):
            cx, cz = side * 2.6, -8.0 + k * (9.0 / max(1, columns // 2 - 1))
            R = 0.32 + 0.025 * np.cos(16.0 * TH) + 0.05 * np.exp(-((YY + 1.7) / 0.25) ** 2) + 0.05 * np.exp(-((YY - 3.9) / 0.25) ** 2)
            P = np.stack([cx + R * np.cos(TH), YY, cz + R * np.sin(TH)], axis=-1)
            N = np.stack([np.cos(TH), np.zeros_like(TH), np.sin(TH)], axis=-1)
            add_grid(P, N, 0 if k % 2 == 0 else 3)
    # arches between consecutive columns of a row: half tori in the x = const planes
    ph = np.linspace(0.0, np.pi, 48 + 1)
    ps = np.linspace(0.0, 2.0 * np.pi, 24 + 1)
    PH, PS = np.meshgrid(ph, ps)
    step = 9.0 / max(1, columns // 2 - 1)
    for side in (-1.0, 1.0):
        for k in range(columns // 2 - 1):
            cx, cz = side * 2.6, -8.0 + (k + 0.5) * step
            Rm, rt = step / 2.0, 0.22
            P = np.stack([cx + rt * np.cos(PS), 4.2 + (Rm + rt * np.sin(PS)) * np.sin(PH), cz + (Rm + rt * np.sin(PS)) * np.cos(PH)], axis=-1)
            N = np.stack([np.cos(PS), np.sin(PS) * np.sin(PH), np.sin(PS) * np.cos(PH)], axis=-1)
            add_grid(P, N, 1)
    # floor (finely tessellated with a gentle ripple), back wall, ceiling strip
    gx = np.linspace(-5.0, 5.0, 200 + 1)
    gz = np.linspace(-10.0, 2.0, 200 + 1)
    GX, GZ = np.meshgrid(gx, gz)
    GY = -2.0 + 0.02 * np.sin(3.0 * GX) * np.sin(2.0 * GZ)
    Nf = np.stack([-0.06 * np.cos(3.0 * GX) * np.sin(2.0 * GZ), np.ones_like(GX), -0.04 * np.sin(3.0 * GX) * np.cos(2.0 * GZ)], axis=-1)
    Nf /= np.linalg.norm(Nf, axis=-1, keepdims=True)
    add_grid(np.stack([GX, GY, GZ], axis=-1), Nf, 2)
    wy = np.linspace(-2.0, 7.5, 60 + 1)
    WX, WY = np.meshgrid(gx, wy)
    add_grid(np.stack([WX, WY, np.full_like(WX, 2.0)], axis=-1), np.tile(np.float64([0, 0, -1]), WX.shape + (1,)), 0)
    add_grid(np.stack([GX[:40], np.full_like(GX[:40], 7.5), GZ[:40] * 0.3 - 1.0], axis=-1), np.tile(np.float64([0, -1, 0]), GX[:40].shape + (1,)), 3)
    lp, ln = _light_quad(-1.2, 1.2, 7.2, -6.0, -4.0)
    mats = _materials([(0.8, 0.78, 0.7), (0.75, 0.45, 0.3), (0.55, 0.55, 0.6), (0.7, 0.7, 0.75)])
    pos = np.concatenate(parts_p + [lp])
    nrm = np.concatenate(parts_n + [ln])
    mi = np.concatenate(parts_m + [np.full(2, len(mats) - 1, np.int32)])
    return build_from_triangles(pos, nrm, mi, mats, camera or camera_bytes(0.0, 2.5, -50.0), bvh)
