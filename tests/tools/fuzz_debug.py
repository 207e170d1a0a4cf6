#!/usr/bin/env python3
"""Re-runs one seed of tests/test_gpu_edge_cases.py::test_fuzz_random_scenes and prints where HIP and oracle differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

seed = int(sys.argv[1])
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
m["dissolve"] = np.where(rng.uniform(0, 1, k) < 0.25, 0.25, 1.0)
m[k - 1]["emission"] = (1, 1, 1)
m[k - 1]["dissolve"] = 1.0
mi = rng.integers(0, k, n).astype(np.int32)
if n > 1:
    mi[0] = 0
s = sc.build_from_triangles(pos, nrm, mi, m).validate()
cam = sc.camera_bytes(float(rng.uniform(-1, 1)), float(rng.uniform(1.5, 3.5)), float(rng.uniform(-60, -20)),
                      float(rng.uniform(-0.03, 0.03)), 0.0, 0.0, int(rng.integers(0, 100)))
W, H = int(rng.integers(1, 70)), int(rng.integers(1, 50))
print("seed", seed, "n", n, "size", size, "k", k, "W,H", W, H, "height", s.height, "materials dissolve", m["dissolve"], "ior", m["ior"])
r = RendererHIP(0)
for prog, path in (("basic", "basic.cl"), ("accumulator", "accumulator.cl"), ("global_illumination", "examples/global_illumination/resources/kernels/global_illumination.cl")):
    for mode in (0, 1):
        got = np.empty((H, W, 3), dtype=np.float32)
        r.render(RenderPropertiesHIP(path, (W, H, 3), got, s, pCamera=cam, kernelMode=mode))
        want = po.render(s, cam, W, H, po.PROGRAMS[prog], mode)
        bad = np.argwhere((got != want).any(axis=2))
        print(prog, "mode", mode, "pixels differing:", len(bad))
        if len(bad):
            cnt = np.zeros((H, W, 4), dtype=np.float32)
            r.render(RenderPropertiesHIP(path, (W, H, 4), cnt, s, pCamera=cam, kernelMode=mode, pixelCounters=True))
            oc = po.pixel_counters(s, cam, W, H, po.PROGRAMS[prog], mode)
            for y, x in bad[:8]:
                print("  y=%d x=%d hip=%s oracle=%s | counters hip=%s oracle=%s" % (y, x, got[y, x], want[y, x], cnt[y, x], oc[y, x]))
