"""Scene buffers in the reference's wire layout (what the renderer plugin uploads,
/root/reference/src/opencl/renderer_opencl.cpp:107-120):

  nodes      LinearBVHNode[M]  32 B  (include/lens_trace/acceleration_structure_explicit.h:20-32)
  prims      Primitive[N]      76 B  (:34-42), BVH-ordered
  materials  Material[K]       32 B  (include/lens_trace/model.h:26-31)
  lights     LightContainer    260 B (acceleration_structure_explicit.h:44-47)
  camera     7 x 4 B                 (src/camera.cpp:14-19)

"LTSB" files are a trivial container for those five raw buffers (tests/golden/*.ltsb are dumps made
by the reference's own host classes, see oracle/ref_host_dump.cpp)."""
import struct
from dataclasses import dataclass, field

import numpy as np

NODE_DTYPE = np.dtype([("boundsMin", "<f4", 3), ("boundsMax", "<f4", 3), ("offset", "<i4"),
                       ("primitiveCount", "<u2"), ("axis", "u1"), ("pad", "u1")])
PRIM_DTYPE = np.dtype([("positionA", "<f4", 3), ("positionB", "<f4", 3), ("positionC", "<f4", 3),
                       ("normalA", "<f4", 3), ("normalB", "<f4", 3), ("normalC", "<f4", 3), ("materialIndex", "<i4")])
MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 3), ("ior", "<f4"), ("dissolve", "<f4"), ("emission", "<f4", 3)])
LIGHT_DTYPE = np.dtype([("count", "<u4"), ("primitives", "<u4", 64)])
assert NODE_DTYPE.itemsize == 32 and PRIM_DTYPE.itemsize == 76 and MATERIAL_DTYPE.itemsize == 32
assert LIGHT_DTYPE.itemsize == 260

_MAGIC = 0x4253544C  # "LTSB"


def camera_bytes(x, y, z, yaw=0.0, pitch=0.0, roll=0.0, frame_count=0):
    """The 28-byte camera buffer: frameCount is a uint bit-copied into the 7th float slot."""
    return struct.pack("<6fI", x, y, z, yaw, pitch, roll, frame_count)


def camera_with_frame(camera28, frame_count):
    return bytes(camera28[:24]) + struct.pack("<I", frame_count)


@dataclass
class Scene:
    nodes: np.ndarray       # uint8, 32*M
    prims: np.ndarray       # uint8, 76*N
    materials: np.ndarray   # uint8, 32*K
    lights: np.ndarray      # uint8, 260
    camera: bytes = field(default_factory=lambda: camera_bytes(0.0, 2.5, -50.0))

    @property
    def node_view(self):
        return self.nodes.view(NODE_DTYPE)

    @property
    def prim_view(self):
        return self.prims.view(PRIM_DTYPE)

    @property
    def material_view(self):
        return self.materials.view(MATERIAL_DTYPE)

    @property
    def light_view(self):
        return self.lights.view(LIGHT_DTYPE)

    @property
    def n_nodes(self):
        return self.nodes.size // 32

    @property
    def n_prims(self):
        return self.prims.size // 76

    def validate(self):
        """Host-side shape checks before anything reaches a kernel (indices in range, sizes whole)."""
        if self.nodes.size % 32 or self.prims.size % 76 or self.materials.size % 32 or self.lights.size != 260:
            raise ValueError("scene buffer sizes are not whole multiples of the reference structs")
        nv, pv = self.node_view, self.prim_view
        if self.n_nodes == 0 or self.n_prims == 0:
            raise ValueError("empty scene")
        leaf = nv["primitiveCount"] > 0
        if leaf.any() and (nv["offset"][leaf].min() < 0 or nv["offset"][leaf].max() >= self.n_prims):
            raise ValueError("leaf primitivesOffset out of range")
        inner = ~leaf
        if inner.any():
            if nv["offset"][inner].min() < 1 or nv["offset"][inner].max() >= self.n_nodes:
                raise ValueError("secondChildOffset out of range")
            if nv["axis"][inner].max() > 2:
                raise ValueError("split axis out of range")
            if np.flatnonzero(inner).max() + 1 >= self.n_nodes:
                raise ValueError("interior node without a left child")
        k = self.materials.size // 32
        if pv["materialIndex"].min() < 0 or pv["materialIndex"].max() >= k:
            raise ValueError("materialIndex out of range")
        lv = self.light_view[0]
        if lv["count"] > 64 or (lv["count"] and lv["primitives"][: lv["count"]].max() >= self.n_prims):
            raise ValueError("light list out of range")
        return self


def load_ltsb(path):
    with open(path, "rb") as f:
        blob = f.read()
    magic, version = struct.unpack_from("<II", blob, 0)
    if magic != _MAGIC or version != 1:
        raise ValueError("%s: not an LTSB v1 file" % path)
    sizes = struct.unpack_from("<5Q", blob, 8)
    off = 48
    parts = []
    for s in sizes:
        parts.append(np.frombuffer(blob, dtype=np.uint8, count=s, offset=off).copy())
        off += s
    return Scene(parts[0], parts[1], parts[2], parts[3], parts[4].tobytes())


def save_ltsb(path, scene):
    cam = np.frombuffer(scene.camera, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", _MAGIC, 1))
        f.write(struct.pack("<5Q", scene.nodes.size, scene.prims.size, scene.materials.size, scene.lights.size, cam.size))
        for a in (scene.nodes, scene.prims, scene.materials, scene.lights, cam):
            f.write(a.tobytes())


# ---- scene construction through the host library (liblenstrace.so: own .obj reader + deterministic BVH builder) ----
_host = None


def _host_lib():
    global _host
    if _host is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liblenstrace.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: run __graft_entry__.build()" % path)
        from . import _capi
        _capi.load()          # liblenstrace.so links liblenstrace-hip.so; load it (and torch's HIP runtime) first
        L = ctypes.CDLL(path)
        vp, u64 = ctypes.c_void_p, ctypes.c_uint64
        L.lt_host_scene_from_obj.argtypes = [ctypes.c_char_p]
        L.lt_host_scene_from_obj.restype = vp
        L.lt_host_scene_from_triangles.argtypes = [vp, vp, vp, u64, vp, u64]
        L.lt_host_scene_from_triangles.restype = vp
        L.lt_host_scene_from_obj_ex.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.lt_host_scene_from_obj_ex.restype = vp
        L.lt_host_scene_from_triangles_ex.argtypes = [vp, vp, vp, u64, vp, u64, ctypes.c_int]
        L.lt_host_scene_from_triangles_ex.restype = vp
        L.lt_host_scene_buffer.argtypes = [vp, ctypes.c_int, ctypes.POINTER(u64)]
        L.lt_host_scene_buffer.restype = vp
        L.lt_host_scene_height.argtypes = [vp]
        L.lt_host_scene_height.restype = ctypes.c_int
        L.lt_host_scene_free.argtypes = [vp]
        L.lt_host_scene_free.restype = None
        _host = L
    return _host


def _scene_from_handle(L, h, camera):
    import ctypes
    if not h:
        raise ValueError("scene construction failed (see the message printed by Model::checkError)")
    try:
        bufs = []
        for which in range(4):
            n = ctypes.c_uint64(0)
            p = L.lt_host_scene_buffer(h, which, ctypes.byref(n))
            bufs.append(np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value,)).copy())
        s = Scene(bufs[0], bufs[1], bufs[2], bufs[3], camera or camera_bytes(0.0, 2.5, -50.0))
        s.height = L.lt_host_scene_height(h)
        return s
    finally:
        L.lt_host_scene_free(h)


# AccelerationStructureExplicitType (include/lens_trace/hip/lens_trace_api.h): the reference's median split, or binned SAH
BVH_MEDIAN, BVH_SAH = 0, 1
# default split rule of build_from_triangles / load_obj and of the synthetic scenes built on them (bench.py --bvh sets it)
default_bvh = BVH_MEDIAN


def load_obj(path, camera=None, bvh=None):
    """Model(path) + AccelerationStructureExplicit, in this repository's own implementation."""
    L = _host_lib()
    return _scene_from_handle(L, L.lt_host_scene_from_obj_ex(str(path).encode(), default_bvh if bvh is None else bvh), camera)


def build_from_triangles(positions, normals, material_indices, materials, camera=None, bvh=None):
    """positions, normals: float32 [N,3,3]; material_indices: int32 [N]; materials: MATERIAL_DTYPE [K]."""
    import ctypes
    L = _host_lib()
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 9)
    nrm = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 9)
    mi = np.ascontiguousarray(material_indices, dtype=np.int32)
    mats = np.ascontiguousarray(materials).view(np.uint8)
    if pos.shape != nrm.shape or mi.shape[0] != pos.shape[0]:
        raise ValueError("triangle arrays disagree in length")
    vp = ctypes.c_void_p
    h = L.lt_host_scene_from_triangles_ex(pos.ctypes.data_as(vp), nrm.ctypes.data_as(vp), mi.ctypes.data_as(vp), pos.shape[0],
                                          mats.ctypes.data_as(vp), mats.size // 32, default_bvh if bvh is None else bvh)
    return _scene_from_handle(L, h, camera)
