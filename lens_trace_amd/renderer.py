"""Python mirror of the renderer plugin surface for this path
(/root/reference/include/lens_trace/renderer.h:5-9, structures.h:51-79, src/opencl/renderer_opencl.cpp:56-153):
RendererHIP.render(RenderPropertiesHIP) fills a caller-owned float buffer, synchronously.  Everything goes
through the C ABI of liblenstrace-hip.so; there is no other compute path."""
import ctypes
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _capi as C
from .scene import Scene

KERNEL_MODE_LINEAR, KERNEL_MODE_TILE = C.KERNEL_MODE_LINEAR, C.KERNEL_MODE_TILE
THREAD_ORGANIZATION_MODE_MAX_FIT, THREAD_ORGANIZATION_MODE_CUSTOM = 0, 1


@dataclass
class ThreadOrganizationHIP:
    """Accepted for API parity with ThreadOrganizationCUDA (structures.h:45-49); pixels never depend on the
    launch decomposition (the reference's CustomBlockSize test), and this backend picks its own."""
    blockSize: tuple = (0, 0)


@dataclass
class RenderPropertiesHIP:
    kernelFilePath: str
    imageDimensions: tuple                       # (W, H, depth)
    pOutputBuffer: np.ndarray                    # float32, >= W*H*depth elements, caller-owned
    pAccelerationStructureExplicit: Scene        # provides node / primitive / light buffers
    pModel: Optional[Scene] = None               # provides the material buffer (defaults to the same Scene)
    pCamera: bytes = b""                         # 28-byte camera buffer
    kernelMode: int = KERNEL_MODE_LINEAR
    threadOrganizationMode: int = THREAD_ORGANIZATION_MODE_MAX_FIT
    threadOrganization: ThreadOrganizationHIP = field(default_factory=ThreadOrganizationHIP)
    # extensions (the reference's unused pNext slot): progressive rendering on the device
    frameFirst: int = 0
    frameCount: int = 0
    accumulate: bool = False
    accumulateBase: int = 0
    giMaxDepth: int = 0
    collectStats: bool = False
    pixelCounters: bool = False                  # diagnostic: per-pixel work counters instead of colour (depth >= 4)
    # Floating-point flavour.  Default: the reference's kernel files as RendererOpenCL builds them on this
    # GPU (clBuildProgram with NULL options) -- bit-identical to them.  strictMath: the same kernels built with
    # -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt.  portableMath: strict, with the device library's approximate
    # leaf functions (rsqrt, sqrt, sinf, cosf, clamp) in correctly rounded forms -- what the CPU oracle computes.
    portableMath: bool = False
    strictMath: bool = False
    # 0: every render() hands the scene buffers to lt_hip_set_scene, which hashes them in full and uploads only when the
    # content changed (the reference uploads on every call).  != 0: the caller versions its scene; the buffers are looked at
    # again only when the objects or this number change.
    sceneVersion: int = 0


def make_desc(program, W, H, depth, camera28, kernel_mode=KERNEL_MODE_LINEAR, frame_first=0, frame_count=0,
              accumulate=False, accumulate_base=0, tile=None, gi_max_depth=0, stats=False, pixel_counters=False, portable_math=False, strict_math=False):
    d = C.RenderDesc()
    d.struct_size = ctypes.sizeof(C.RenderDesc)
    d.program, d.kernel_mode = program, kernel_mode
    d.width, d.height, d.depth = W, H, depth
    cam = bytes(camera28)
    if len(cam) != 28:
        raise ValueError("camera buffer must be 28 bytes")
    ctypes.memmove(d.camera, cam, 28)
    d.frame_first, d.frame_count = frame_first, frame_count
    d.accumulate, d.accumulate_base = int(bool(accumulate)), accumulate_base
    if tile is not None:
        d.tile_w, d.tile_h, d.tile_first, d.tile_stride = tile
    d.gi_max_depth = gi_max_depth
    d.flags = ((C.RENDER_FLAG_STATS if stats else 0) | (C.RENDER_FLAG_PIXEL_COUNTERS if pixel_counters else 0) |
               (C.RENDER_FLAG_PORTABLE_MATH if portable_math else 0) | (C.RENDER_FLAG_STRICT_MATH if strict_math else 0))
    return d


class RendererHIP:
    """One context per GPU.  `device` is the HIP ordinal."""

    def __init__(self, device=0):
        self._L = C.load()
        self._ctx = ctypes.c_void_p()
        rc = self._L.lt_hip_create(device, ctypes.byref(self._ctx))
        if rc:
            raise C.LensTraceError(rc, self._L.lt_hip_last_error(None).decode())
        self._scene_key = None
        self._scene_refs = None
        self._scene_version = 0

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.lt_hip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise C.LensTraceError(rc, self._L.lt_hip_last_error(self._ctx).decode())

    # -- scene -------------------------------------------------------------------------------------
    def set_scene(self, scene: Scene, materials: Optional[Scene] = None):
        m = (materials or scene).materials
        arrs = [np.ascontiguousarray(a) for a in (scene.nodes, scene.prims, m, scene.lights)]
        args = []
        for a in arrs:
            args += [a.ctypes.data_as(ctypes.c_void_p), a.nbytes]
        self._check(self._L.lt_hip_set_scene(self._ctx, *args))
        # keep the objects alive so that their ids cannot be recycled by a new scene
        self._scene_refs = (scene, materials or scene)
        self._scene_key = (id(scene), id(materials or scene))

    def invalidate_scene(self):
        """Forget the uploaded scene (call after modifying scene buffers in place)."""
        self._scene_key = None
        self._scene_refs = None

    def resolve_program(self, kernel_file_path):
        """Built-in program by basename, or a user .hip file compiled with hipRTC at first use and cached by path."""
        out = ctypes.c_int(0)
        self._check(self._L.lt_hip_resolve_program(self._ctx, str(kernel_file_path).encode(), ctypes.byref(out)))
        return out.value

    # -- the plugin entry point ----------------------------------------------------------------------
    def render(self, props: RenderPropertiesHIP):
        W, H, D = props.imageDimensions
        out = props.pOutputBuffer
        if out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("pOutputBuffer must be contiguous float32")
        a = props.pAccelerationStructureExplicit
        m = props.pModel or a
        program = self.resolve_program(props.kernelFilePath)
        d = make_desc(program, W, H, D, props.pCamera, props.kernelMode, props.frameFirst, props.frameCount,
                      props.accumulate, props.accumulateBase, None, props.giMaxDepth, props.collectStats, props.pixelCounters, props.portableMath, props.strictMath)
        # the reference uploads its scene on every call; a caller that versions its scene has it looked at only when the objects
        # or the number change, anyone else hands it over with the frame (lt_hip_render_scene: hashed in full while the frame
        # renders, uploaded -- and the frame rendered again -- only when a byte changed)
        key = (id(a), id(m))
        if props.sceneVersion and key == self._scene_key and props.sceneVersion == self._scene_version:
            self._check(self._L.lt_hip_render(self._ctx, ctypes.byref(d), out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
            return
        arrs = [np.ascontiguousarray(x) for x in (a.nodes, a.prims, m.materials, a.lights)]
        args = []
        for x in arrs:
            args += [x.ctypes.data_as(ctypes.c_void_p), x.nbytes]
        self._check(self._L.lt_hip_render_scene(self._ctx, *args, ctypes.byref(d), out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        self._scene_refs = (a, m)
        self._scene_key = key
        self._scene_version = props.sceneVersion

    # -- device-resident variants (bench / multi-GPU) -----------------------------------------------------
    def output_floats(self, desc):
        n = ctypes.c_uint64(0)
        self._check(self._L.lt_hip_output_floats(ctypes.byref(desc), ctypes.byref(n)))
        return n.value

    def render_device(self, desc, out_ptr, out_bytes, stream=0):
        self._check(self._L.lt_hip_render_device(self._ctx, ctypes.byref(desc), ctypes.c_void_p(out_ptr), out_bytes,
                                                 ctypes.c_void_p(stream)))

    def untile(self, gathered_ptr, floats_per_rank, n_ranks, W, H, D, tile_w, tile_h, image_ptr, stream=0):
        self._check(self._L.lt_hip_untile(self._ctx, ctypes.c_void_p(gathered_ptr), floats_per_rank, n_ranks, W, H, D,
                                          tile_w, tile_h, ctypes.c_void_p(image_ptr), ctypes.c_void_p(stream)))

    def synchronize(self, stream=0):
        self._check(self._L.lt_hip_synchronize(self._ctx, ctypes.c_void_p(stream)))

    def scene_structure(self, what):
        """lt_hip_read_scene_structure: 0 own tree (NODE_DTYPE array), 1 leaf order table (n_prims x 8 uint32), 2 the per-lane walks'
        array (bytes), 3 (own height, group-tree height, groups, prepared on the device).  None when the scene has no such structure."""
        from . import scene as sc
        n = ctypes.c_uint64(0)
        self._check(self._L.lt_hip_read_scene_structure(self._ctx, what, None, ctypes.c_uint64(0), ctypes.byref(n)))
        if n.value == 0:
            return None
        buf = np.zeros(n.value, dtype=np.uint8)
        self._check(self._L.lt_hip_read_scene_structure(self._ctx, what, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(buf.nbytes), ctypes.byref(n)))
        if what == 0:
            return buf.view(sc.NODE_DTYPE)
        if what == 1:
            return buf.view(np.uint32).reshape(-1, 8)
        if what == 3:
            return tuple(int(x) for x in buf.view(np.int32))
        return buf

    def stats(self):
        s = C.Stats()
        self._check(self._L.lt_hip_get_stats(self._ctx, ctypes.byref(s)))
        return s.as_dict()
