#!/usr/bin/env python3
"""Soak of the device-side scene preparation against the host's (tests/test_gpu_device_prep.py's comparison, many more scenes):
clustered and uniform random triangle sets of 2 .. 400 000 triangles, height slack 0 / 1 / 2 (no slack: the median rule and the
demotion of large ranges to the looping wavefronts), coincident centroids.  Prints one line per scene; exits non-zero on the first
difference.   usage: python tests/tools/prep_soak.py [scenes]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lens_trace_amd import scene as sc  # noqa: E402
from lens_trace_amd.renderer import RendererHIP  # noqa: E402


def scene_of(rng, n, clusters, coincident):
    if clusters:
        cc = rng.uniform(-8, 8, (clusters, 3))
        centre = cc[rng.integers(0, clusters, n)] + rng.normal(0, rng.choice([0.001, 0.05, 0.5]), (n, 3))
    else:
        centre = rng.uniform(-4, 4, (n, 3))
    if coincident:
        centre[:coincident] = centre[0]
    pos = (centre[:, None, :] + rng.normal(0, rng.choice([0.0005, 0.02, 0.3]), (n, 3, 3))).astype(np.float32)
    nrm = np.tile(np.float32([0, 0, -1]), (n, 3, 1))
    m = np.zeros(3, dtype=sc.MATERIAL_DTYPE)
    m["diffuse"], m["ior"], m["dissolve"] = 0.5, 1.3, 1.0
    m[2]["emission"] = (1, 1, 1)
    mi = rng.integers(0, 2, n).astype(np.int32)
    mi[0] = 2
    return sc.build_from_triangles(pos, nrm, mi, m).validate()


def structures(scene, device, slack):
    os.environ["LT_DEVICE_BUILD"] = "1" if device else "0"
    os.environ["LT_RETREE_SLACK"] = str(slack)
    r = RendererHIP(0)
    r.set_scene(scene)
    got = [r.scene_structure(k) for k in range(4)]
    r.close()
    return got


total = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(os.environ.get("LT_SOAK_SEED", "2026")))
for k in range(total):
    n = int(rng.choice([2, 7, 63, 64, 65, 200, 2047, 2049, 5000, 30000, 120000, 400000], p=[.04, .04, .04, .04, .04, .1, .1, .1, .2, .15, .1, .05]))
    clusters = int(rng.choice([0, 1, 3, 40]))
    coincident = int(rng.choice([0, 0, n // 3, n])) if n < 40000 else 0
    slack = int(rng.choice([0, 1, 2]))
    s = scene_of(rng, n, clusters, coincident)
    host, dev = structures(s, False, slack), structures(s, True, slack)
    ok = host[3][:3] == dev[3][:3] and dev[3][3] == 1 and all(
        (a is None and b is None) or (a is not None and b is not None and np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)))
        for a, b in zip(host[:3], dev[:3]))
    print("scene %3d: %6d triangles, %2d clusters, %6d coincident, slack %d: heights %s  %s" % (k, n, clusters, coincident, slack, dev[3][:2], "same" if ok else "DIFFERENT"), flush=True)
    if not ok:
        sys.exit(1)
print("all %d scenes: device-side and host-side preparation byte for byte the same" % total)
