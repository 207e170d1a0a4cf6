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


# The HIP path's default floating-point flavour is the reference kernels' own on this GPU (device-library rsqrt / sqrt /
# sinf / cosf / clamp); the CPU oracle computes the portable, correctly rounded flavour.  Tests that compare with the oracle
# (or with golden pixels made by it) import these two in place of RenderPropertiesHIP / make_desc.
def oracle_props(*a, **kw):
    from lens_trace_amd.renderer import RenderPropertiesHIP
    kw.setdefault("portableMath", True)
    return RenderPropertiesHIP(*a, **kw)


def oracle_desc(*a, **kw):
    from lens_trace_amd.renderer import make_desc
    kw.setdefault("portable_math", True)
    return make_desc(*a, **kw)
