import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_index():
    rows = []
    with open(os.path.join(GOLDEN, "index.txt")) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            t = line.split()
            rows.append(dict(tag=t[0], scene=t[1], program=t[2], mode=int(t[3]), W=int(t[4]), H=int(t[5]),
                             frame=int(t[6]), yaw=float(t[7]), rays=int(t[8]), shadow_rays=int(t[9]),
                             node_visits=int(t[10]), tri_tests=int(t[11])))
    return rows


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The HIP path's default floating-point flavour is the reference kernels' own on this GPU, as the reference builds them (NULL
# OpenCL build options: contracted a*b+c, device-library rsqrt / sqrt / sinf / cosf / divide); the CPU oracle computes the
# portable, correctly rounded flavour.  Tests that compare with the oracle (or with golden pixels made by it) import these two
# in place of RenderPropertiesHIP / make_desc.
def oracle_props(*a, **kw):
    from lens_trace_amd.renderer import RenderPropertiesHIP
    kw.setdefault("portableMath", True)
    return RenderPropertiesHIP(*a, **kw)


def oracle_desc(*a, **kw):
    from lens_trace_amd.renderer import make_desc
    kw.setdefault("portable_math", True)
    return make_desc(*a, **kw)


def fuzz_scene(seed):
    """One seeded random small scene (1..400 triangles of random size and orientation, 2..5 materials with lens materials and an
    emissive one), camera and image size; returns (scene, camera bytes, W, H, rng) -- the rng continues the same stream."""
    import numpy as np
    from lens_trace_amd import scene as sc
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 400))
    centre = np.stack([rng.uniform(-4, 4, n), rng.uniform(-1.5, 6.5, n), rng.uniform(-6, 1, n)], axis=-1)
    size = 10.0 ** rng.uniform(-1.5, 0.3)
    pos = (centre[:, None, :] + rng.normal(0, size, (n, 3, 3))).astype(np.float32)
    nrm = rng.normal(0, 1, (n, 3, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    k = int(rng.integers(2, 6))
    m = np.zeros(k, dtype=sc.MATERIAL_DTYPE)
    m["diffuse"] = rng.uniform(0, 1, (k, 3))
    m["ior"] = rng.uniform(1.0, 2.0, k)
    m["dissolve"] = np.where(rng.uniform(0, 1, k) < 0.25, 0.25, 1.0)     # some lens materials (basic's refraction path)
    m[k - 1]["emission"] = (1, 1, 1)
    m[k - 1]["dissolve"] = 1.0
    mi = rng.integers(0, k, n).astype(np.int32)
    if n > 1:
        mi[0] = 0                                                           # keep the first triangle non-emissive most of the time
    s = sc.build_from_triangles(pos, nrm, mi, m).validate()
    cam = sc.camera_bytes(float(rng.uniform(-1, 1)), float(rng.uniform(1.5, 3.5)), float(rng.uniform(-60, -20)),
                          float(rng.uniform(-0.03, 0.03)), 0.0, 0.0, int(rng.integers(0, 100)))
    W, H = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    return s, cam, W, H, rng
