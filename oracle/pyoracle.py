"""ctypes binding of the CPU oracle (oracle/lt_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); never by
lens_trace_amd (the product)."""
import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblt_oracle.so")

BASIC, BASIC_LIGHTING, ACCUMULATOR, GI, GI25, CUSTOM = range(6)
PROGRAMS = {"basic": BASIC, "basic_lighting": BASIC_LIGHTING, "accumulator": ACCUMULATOR,
            "global_illumination": GI, "global_illumination25": GI25, "custom_opencl": CUSTOM}
MODE_LINEAR, MODE_TILE = 0, 1


class Stats(ctypes.Structure):
    _fields_ = [("rays", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64), ("node_visits", ctypes.c_uint64),
                ("tri_tests", ctypes.c_uint64), ("max_stack", ctypes.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "lt_oracle.c")):
            build()
        L = ctypes.CDLL(_SO)
        vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        L.lt_oracle_render.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, u32, i32, ctypes.POINTER(Stats)]
        L.lt_oracle_render.restype = i32
        L.lt_oracle_render_opencl_launch.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, u32, i32,
                                                     ctypes.POINTER(Stats)]
        L.lt_oracle_render_opencl_launch.restype = i32
        L.lt_oracle_pixel_counters.argtypes = [i32, i32, vp, vp, vp, vp, vp, u32, u32, i32, vp]
        L.lt_oracle_pixel_counters.restype = i32
        L.lt_oracle_accumulate.argtypes = [vp, vp, ctypes.c_uint64, u32]
        L.lt_oracle_accumulate.restype = None
        L.lt_oracle_random.argtypes = [ctypes.c_float] * 3
        L.lt_oracle_random.restype = ctypes.c_float
        L.lt_oracle_trace.argtypes = [i32, vp, vp, vp, vp, ctypes.c_float, i32, i32, ctypes.POINTER(i32), vp]
        L.lt_oracle_trace.restype = i32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _scene_arrays(scene):
    """scene: object with .nodes .prims .materials .lights (numpy uint8 raw buffers)."""
    return [np.ascontiguousarray(getattr(scene, k)) for k in ("nodes", "prims", "materials", "lights")]


def render(scene, camera28, W, H, program, mode=MODE_LINEAR, depth=3, gi_max_depth=16, threads=1, rows=None,
           want_stats=False):
    """Whole image (every pixel once).  camera28: 28 raw bytes.  Returns float32 [H,W,depth] (and stats)."""
    L = lib()
    n, p, m, l = _scene_arrays(scene)
    cam = np.frombuffer(bytes(camera28), dtype=np.uint8).copy()
    out = np.zeros((H, W, depth), dtype=np.float32)
    y0, y1 = rows if rows is not None else (0, H)
    threads = max(1, min(threads, y1 - y0))
    # interleaved small bands: per-row cost is very uneven
    band = max(1, min(16, (y1 - y0) // (threads * 4) or 1))
    bands = [(y, min(y + band, y1)) for y in range(y0, y1, band)]
    stats = [Stats() for _ in bands]

    def run(i):
        a, b = bands[i]
        rc = L.lt_oracle_render(program, mode, _p(n), _p(p), _p(m), _p(l), _p(cam), _p(out), W, H, depth, a, b,
                                gi_max_depth, ctypes.byref(stats[i]))
        if rc:
            raise RuntimeError("oracle: traversal stack overflow (>64 entries; undefined in the reference)")

    if threads == 1:
        for i in range(len(bands)):
            run(i)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(run, range(len(bands))))
    if want_stats:
        tot = {k: 0 for k, _ in Stats._fields_}
        for s in stats:
            for k, v in s.as_dict().items():
                tot[k] = max(tot[k], v) if k == "max_stack" else tot[k] + v
        return out, tot
    return out


def render_opencl_launch(scene, camera28, W, H, program, mode, global_size, local_size, depth=3, gi_max_depth=16,
                         fill=np.nan):
    """The OpenCL backend's work-block decomposition (renderer_opencl.cpp:80-146)."""
    L = lib()
    n, p, m, l = _scene_arrays(scene)
    cam = np.frombuffer(bytes(camera28), dtype=np.uint8).copy()
    out = np.full((H, W, depth), fill, dtype=np.float32)
    st = Stats()
    rc = L.lt_oracle_render_opencl_launch(program, mode, _p(n), _p(p), _p(m), _p(l), _p(cam), _p(out), W, H, depth,
                                          global_size[0], global_size[1], local_size[0], local_size[1], gi_max_depth,
                                          ctypes.byref(st))
    if rc:
        raise RuntimeError("oracle launch error %d" % rc)
    return out


def pixel_counters(scene, camera28, W, H, program, mode=MODE_LINEAR, gi_max_depth=16):
    """uint32 [H,W,4] = rays, shadow rays, node visits, triangle tests per pixel."""
    L = lib()
    n, p, m, l = _scene_arrays(scene)
    cam = np.frombuffer(bytes(camera28), dtype=np.uint8).copy()
    out = np.zeros((H, W, 4), dtype=np.uint32)
    if L.lt_oracle_pixel_counters(program, mode, _p(n), _p(p), _p(m), _p(l), _p(cam), W, H, gi_max_depth, _p(out)):
        raise RuntimeError("oracle: traversal stack overflow")
    return out


def accumulate(acc, frame, n):
    L = lib()
    assert acc.dtype == np.float32 and frame.dtype == np.float32 and acc.flags.c_contiguous and frame.flags.c_contiguous
    L.lt_oracle_accumulate(_p(acc), _p(frame), acc.size, n)


def random(u, v, seed):
    return float(lib().lt_oracle_random(u, v, seed))


def trace(scene, origin, direction, program=ACCUMULATOR, tmax=np.finfo(np.float32).max, ignore=None):
    L = lib()
    n, p, _, _ = _scene_arrays(scene)
    o = np.asarray(origin, dtype=np.float32)
    d = np.asarray(direction, dtype=np.float32)
    prim = ctypes.c_int(0)
    tuv = np.zeros(3, dtype=np.float32)
    hit = L.lt_oracle_trace(program, _p(n), _p(p), _p(o), _p(d), tmax, 0 if ignore is None else 1,
                            0 if ignore is None else ignore, ctypes.byref(prim), _p(tuv))
    return hit, prim.value, tuv
