#!/usr/bin/env python3
"""Diagnostic: the reference-semantics entry point (lt_hip_render: caller-owned HOST output buffer) at the bench workload:
kernel time vs wall time including the staging memset and the 99.5 MB read-back over PCIe."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402

s = synth.heightfield_wall(708)
r = RendererHIP(0)
W, H = 3840, 2160
out = np.empty((H, W, 3), dtype=np.float32)
p = RenderPropertiesHIP("accumulator.cl", (W, H, 3), out, s, pCamera=s.camera, frameFirst=1, frameCount=16, accumulate=True)
r.render(p)
for _ in range(3):
    t0 = time.perf_counter()
    r.render(p)
    wall = (time.perf_counter() - t0) * 1e3
    st = r.stats()
    print("16-sample 4K frame into a host buffer: kernels %.1f ms, lt_hip_render total %.1f ms, python wall %.1f ms" % (st["kernel_ms"], st["total_ms"], wall))
