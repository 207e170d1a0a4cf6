"""ctypes binding of liblenstrace-hip.so (include/lenstrace_hip.h).  There is NO fallback: if the library is
missing or a call fails, this raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LT_HIP_LIBRARY") or os.path.join(_HERE, "lib", "liblenstrace-hip.so")

(LT_OK, LT_ERR_INVALID_ARGUMENT, LT_ERR_NO_DEVICE, LT_ERR_HIP, LT_ERR_NO_SCENE, LT_ERR_BAD_SCENE,
 LT_ERR_BUFFER_TOO_SMALL, LT_ERR_UNKNOWN_PROGRAM) = range(8)
STATUS_NAMES = ["LT_OK", "LT_ERR_INVALID_ARGUMENT", "LT_ERR_NO_DEVICE", "LT_ERR_HIP", "LT_ERR_NO_SCENE",
                "LT_ERR_BAD_SCENE", "LT_ERR_BUFFER_TOO_SMALL", "LT_ERR_UNKNOWN_PROGRAM"]

PROGRAM_BASIC, PROGRAM_BASIC_LIGHTING, PROGRAM_ACCUMULATOR, PROGRAM_GLOBAL_ILLUMINATION, PROGRAM_GLOBAL_ILLUMINATION_25 = range(5)
PROGRAM_CUSTOM_OPENCL = 5
KERNEL_MODE_LINEAR, KERNEL_MODE_TILE = 0, 1
RENDER_FLAG_STATS = 1
RENDER_FLAG_PIXEL_COUNTERS = 2
RENDER_FLAG_DEVICE_LIBM = 4      # ignored (ABI 2 name of RENDER_FLAG_STRICT_MATH)
RENDER_FLAG_PORTABLE_MATH = 8
RENDER_FLAG_STRICT_MATH = 16
RENDER_FLAG_NO_WALK_TIMING = 32

# every symbol include/lenstrace_hip.h declares
EXPORTS = ["lt_hip_abi_version", "lt_hip_create", "lt_hip_destroy", "lt_hip_last_error", "lt_hip_program_from_path",
           "lt_hip_resolve_program",
           "lt_hip_set_scene", "lt_hip_output_floats", "lt_hip_render", "lt_hip_render_scene", "lt_hip_render_device", "lt_hip_untile",
           "lt_hip_synchronize", "lt_hip_get_stats", "lt_hip_own_hierarchy", "lt_hip_own_wide", "lt_hip_read_scene_structure"]


class RenderDesc(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("program", ctypes.c_int32), ("kernel_mode", ctypes.c_int32),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("depth", ctypes.c_uint32),
                ("camera", ctypes.c_uint8 * 28),
                ("frame_first", ctypes.c_uint32), ("frame_count", ctypes.c_uint32), ("accumulate", ctypes.c_uint32),
                ("accumulate_base", ctypes.c_uint32),
                ("tile_w", ctypes.c_uint32), ("tile_h", ctypes.c_uint32), ("tile_first", ctypes.c_uint32),
                ("tile_stride", ctypes.c_uint32),
                ("gi_max_depth", ctypes.c_int32), ("flags", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    _fields_ = [("rays", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64), ("node_visits", ctypes.c_uint64),
                ("tri_tests", ctypes.c_uint64), ("pixels", ctypes.c_uint64), ("frames", ctypes.c_uint32),
                ("kernel_launches", ctypes.c_uint32), ("kernel_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("render_ms", ctypes.c_float), ("shadow_packets", ctypes.c_int32),
                ("scene_uploads", ctypes.c_uint32), ("scene_reused", ctypes.c_uint32),
                ("own_tree_height", ctypes.c_int32), ("own_tree_ms", ctypes.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LensTraceError(RuntimeError):
    def __init__(self, code, text):
        self.code = code
        name = STATUS_NAMES[code] if 0 <= code < len(STATUS_NAMES) else str(code)
        super().__init__("%s: %s" % (name, text))


_lib = None


def load():
    """Loads liblenstrace-hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as the
    # system one).  Importing torch first makes this library bind to the copy torch uses, so device pointers,
    # streams and RCCL buffers are shared; loaded the other way round, the process ends up with two runtimes
    # and torch reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: the HIP extension is not built (run __graft_entry__.build()); "
                          "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.lt_hip_abi_version.restype = i32
    L.lt_hip_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.lt_hip_destroy.argtypes = [vp]
    L.lt_hip_last_error.argtypes = [vp]
    L.lt_hip_last_error.restype = ctypes.c_char_p
    L.lt_hip_program_from_path.argtypes = [ctypes.c_char_p, ctypes.POINTER(i32)]
    L.lt_hip_resolve_program.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i32)]
    L.lt_hip_set_scene.argtypes = [vp, vp, u64, vp, u64, vp, u64, vp, u64]
    L.lt_hip_output_floats.argtypes = [ctypes.POINTER(RenderDesc), ctypes.POINTER(u64)]
    L.lt_hip_render.argtypes = [vp, ctypes.POINTER(RenderDesc), vp, u64]
    if hasattr(L, "lt_hip_render_scene"):
        L.lt_hip_render_scene.argtypes = [vp, vp, u64, vp, u64, vp, u64, vp, u64, ctypes.POINTER(RenderDesc), vp, u64]
    L.lt_hip_render_device.argtypes = [vp, ctypes.POINTER(RenderDesc), vp, u64, vp]
    L.lt_hip_untile.argtypes = [vp, vp, u64, u32, u32, u32, u32, u32, u32, vp, vp]
    L.lt_hip_synchronize.argtypes = [vp, vp]
    L.lt_hip_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    for name in EXPORTS:
        if not hasattr(L, name) and os.environ.get("LT_HIP_LIBRARY"):
            continue      # (an older build of the library loaded for an A/B measurement, tests/tools/ab_libs.sh)
        if name not in ("lt_hip_last_error",):
            getattr(L, name).restype = i32
    if L.lt_hip_abi_version() != 4:
        raise ImportError("liblenstrace-hip.so ABI version mismatch")
    _lib = L
    return L


def program_from_path(path):
    out = ctypes.c_int(0)
    rc = load().lt_hip_program_from_path(path.encode(), ctypes.byref(out))
    if rc:
        raise LensTraceError(rc, "no built-in program for kernel file %r" % path)
    return out.value


def own_hierarchy(nodes, n_prims=0, height_slack=2, want_ranks=False):
    """lt_hip_own_hierarchy (host only): (height, nodes of the backend's own hierarchy as a NODE_DTYPE array, rank8 or None);
    height -1 and no arrays when the scene gets none."""
    import numpy as np
    from . import scene as sc
    nodes = np.ascontiguousarray(nodes)
    leaves = int((nodes["primitiveCount"] != 0).sum())
    out = np.zeros(max(1, 2 * leaves - 1), dtype=sc.NODE_DTYPE)
    ranks = np.zeros((max(1, n_prims), 8), dtype=np.uint32) if want_ranks else None
    h = load().lt_hip_own_hierarchy(nodes.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(nodes.nbytes), ctypes.c_int(height_slack),
                                    out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(out.nbytes),
                                    ranks.ctypes.data_as(ctypes.c_void_p) if want_ranks else None, ctypes.c_uint32(n_prims))
    if h < 0:
        return -1, None, None
    return h, out, ranks


OWN16_DTYPE = [("q", "<u2", (6,)), ("link", "<u4")]   # a child slot: lo.x lo.y lo.z hi.x hi.y hi.z on the grid; the child's group or 0x80000000 | leaf record


def own_wide(own_nodes, n_prims):
    """lt_hip_own_wide (host only): (height of the group tree, origin[3], step[3], slots as an OWN16_DTYPE array of shape (groups, 4))."""
    import numpy as np
    own_nodes = np.ascontiguousarray(own_nodes)
    frame = np.zeros(6, dtype=np.float32)
    out = np.zeros((max(1, len(own_nodes) // 2), 4), dtype=OWN16_DTYPE)
    assert out.dtype.itemsize == 16
    groups = ctypes.c_uint32(0)
    h = load().lt_hip_own_wide(own_nodes.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(own_nodes.nbytes), ctypes.c_uint32(n_prims),
                               frame.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(out.nbytes),
                               ctypes.byref(groups))
    if h < 0:
        raise LensTraceError(LT_ERR_BAD_SCENE, "lt_hip_own_wide failed")
    return h, frame[:3].copy(), frame[3:].copy(), out[:groups.value].copy()
