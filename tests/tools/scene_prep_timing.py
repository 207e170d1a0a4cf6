"""lt_hip_set_scene of the bench scene (1 M triangles) with the scene prepared on the device and on the host: wall time per call
(fresh content every time: one vertex nudged, so that nothing is taken from the resident copy), and the laps
(LT_DEBUG_SCENE_TIMING=1 prints them to stderr).  Run on the GPU box: python tests/tools/scene_prep_timing.py [cells]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lens_trace_amd import synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 708
s = synth.heightfield_wall(cells).validate()
print("triangles", s.n_prims, "nodes", s.n_nodes)
for device in ("1", "0"):
    os.environ["LT_DEVICE_BUILD"] = device
    r = RendererHIP(0)
    r.set_scene(s)
    times = []
    for k in range(6):
        nodes = s.node_view.copy()
        nodes["boundsMax"][0][0] += 1e-3 * (k + 1)      # the root's box a little larger: a new node buffer, the same tree
        s2 = type(s)(nodes=nodes.view(np.uint8).reshape(-1), prims=s.prims, materials=s.materials, lights=s.lights, camera=s.camera)
        if k == 5:
            os.environ["LT_DEBUG_SCENE_TIMING"] = "1"
        t = time.perf_counter()
        r.set_scene(s2)
        times.append((time.perf_counter() - t) * 1e3)
    os.environ.pop("LT_DEBUG_SCENE_TIMING", None)
    print("LT_DEVICE_BUILD=%s: set_scene of a new node buffer %s ms; structures %s" % (device, " ".join("%.1f" % t for t in times), r.scene_structure(3)))
    r.close()
