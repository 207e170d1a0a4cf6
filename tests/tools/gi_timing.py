#!/usr/bin/env python3
"""Diagnostic: kernel time of the global-illumination program, wavefront pipeline vs the one-lane-per-pixel kernel
(LT_GI_MEGAKERNEL=1), on the Cornell box (1080p) and on the 1 M-triangle wall (4K)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc, synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402

GI = "examples/global_illumination/resources/kernels/global_illumination.cl"
if os.environ.get("LT_GI25") == "1":   # the 25-sample variant (the reference CLI's global_illumination.scene)
    GI = "resources/kernels/opencl/global_illumination.cl"
FRAMES = int(os.environ.get("LT_FRAMES", "1"))   # > 1: a running mean of that many frames per call (ms printed per call)
r = RendererHIP(0)
for name, scene, W, H in (("cornell", sc.load_ltsb(os.path.join(ROOT, "tests", "golden", "cornell_box_O0.ltsb")), 1920, 1080),
                          ("wall-1M", synth.heightfield_wall(708), 3840, 2160)):
    out = np.empty((H, W, 3), dtype=np.float32)
    for depth in (16, 4):
        p = RenderPropertiesHIP(GI, (W, H, 3), out, scene, pCamera=sc.camera_with_frame(scene.camera, 2), giMaxDepth=depth,
                                **({"frameFirst": 1, "frameCount": FRAMES, "accumulate": True} if FRAMES > 1 else {}))
        r.render(p)
        ms = []
        for _ in range(3):
            r.render(p)
            ms.append(r.stats()["kernel_ms"])
        print("%s %dx%d depth %d: %.2f ms (launches %d), mode %s" % (name, W, H, depth, min(ms), r.stats()["kernel_launches"],
                                                                   "megakernel" if os.environ.get("LT_GI_MEGAKERNEL") == "1" else "wavefront"))
