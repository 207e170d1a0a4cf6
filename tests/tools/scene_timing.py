#!/usr/bin/env python3
"""Diagnostic: what lt_hip_set_scene spends on the 1 M-triangle wall (LT_DEBUG_SCENE_TIMING=1 prints the laps), and the cost of
the cheap paths: the same scene again (hash only), a material edited, the primitives edited."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LT_DEBUG_SCENE_TIMING"] = "1"
from lens_trace_amd import scene as sc, synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP  # noqa: E402

s = synth.heightfield_wall(708).validate()
r = RendererHIP(0)


def timed(what):
    t0 = time.perf_counter()
    r.set_scene(s)
    print("%-40s %.2f ms" % (what, (time.perf_counter() - t0) * 1e3), flush=True)


timed("first upload (all builds)")
timed("the same scene again (hash)")
s.materials.view(sc.MATERIAL_DTYPE)["diffuse"][0, 0] += 0.01
timed("a material edited")
s.prims.view(sc.PRIM_DTYPE)["normalA"][::5] *= -1.0
timed("primitives edited (nodes untouched)")
nodes = s.nodes.view(sc.NODE_DTYPE)
leaf = int(np.flatnonzero(nodes["primitiveCount"] != 0)[5])
nodes["boundsMax"][leaf] += 1e-6
timed("a node edited (all builds again)")
