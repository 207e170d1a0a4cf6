import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from lens_trace_amd import scene as sc
from lens_trace_amd.renderer import RendererHIP
from oracle import pyoracle as po
from tests.conftest import oracle_props as Props
from tests.test_gpu_edge_cases import quad_prims, mats, PATHS
prims = quad_prims(8, z0=0.0, dz=0.5)
prims[3]["positionB"] = prims[3]["positionA"]
m = mats([(0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8), (1, 1, 1)], emissive=(3,))
prims[5]["materialIndex"] = 3
s = sc.build_from_triangles(np.stack([prims["positionA"], prims["positionB"], prims["positionC"]], axis=1),
                         np.stack([prims["normalA"], prims["normalB"], prims["normalC"]], axis=1), prims["materialIndex"], m)
for mode in ("0","1","2"):
    os.environ["LT_SHADOW_PACKETS"]=mode
    r = RendererHIP(0)
    for cam in (sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0, 0, 1), sc.camera_bytes(5.0, 7.5, -50.0, 0.0, 0, 0, 2), sc.camera_bytes(-5.0, -2.5, -20.0, 0.0, 0, 0, 2)):
        for prog in ("basic","accumulator","global_illumination"):
            for W in (33,32):
                got = np.empty((W, W, 3), dtype=np.float32)
                r.render(Props(PATHS[prog], (W, W, 3), got, s, pCamera=cam))
                want = po.render(s, cam, W, W, po.PROGRAMS[prog], 0, gi_max_depth=16)
                d = np.argwhere((got!=want).any(axis=2))
                if len(d): print("mode",mode,"cam",np.frombuffer(cam,dtype=np.float32)[:3],prog,W,"diff pixels (y,x):",d.tolist(),[ (got[y,x].tolist(),want[y,x].tolist()) for y,x in d[:3]])
    r.close()
print("done")
