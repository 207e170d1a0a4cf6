#!/usr/bin/env python3
"""Diagnostic: render the bench scenes once with whatever library LT_HIP_LIBRARY names and save the images, so that two
builds can be compared pixel for pixel (python tests/tools/ab_dump.py <tag>; then np.array_equal on the .npy files)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc, synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402

tag = sys.argv[1]
ACC = "examples/accumulator/resources/kernels/accumulator.cl"
GI = "examples/global_illumination/resources/kernels/global_illumination.cl"
r = RendererHIP(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name, scene, W, H, prog in (("wall", synth.heightfield_wall(708), 1920, 1080, ACC), ("blob", synth.blob_in_box(), 1920, 1080, ACC),
                                ("soup", synth.triangle_soup(200000), 1280, 720, ACC), ("colonnade", synth.colonnade(), 1280, 720, ACC),
                                ("cornell_gi", sc.load_ltsb(os.path.join(ROOT, "tests", "golden", "cornell_box_O0.ltsb")), 640, 360, GI)):
    out = np.empty((H, W, 3), dtype=np.float32)
    for frame in (1, 2, 3):
        r.render(RenderPropertiesHIP(prog, (W, H, 3), out, scene, pCamera=sc.camera_with_frame(scene.camera, frame)))
        np.save(os.path.join(ROOT, "gpurun_out", "ab_%s_%s_%d.npy" % (tag, name, frame)), out.astype(np.float16) if False else out)
        print(tag, name, frame, "kernel ms %.3f" % r.stats()["kernel_ms"], "mean %.6f" % out.mean())
