"""Runs the REFERENCE's own OpenCL kernels on the MI355X.  TEST INFRASTRUCTURE ONLY.

oracle/build_ref.sh compiles the reference's .cl files (where they lie under /root/reference) for gfx950 with the
image's ROCm clang and its real OpenCL device libraries into oracle/_ref/<kernel>.{strict,default}.co -- the code
object clBuildProgram would produce on this GPU.  This module loads such a code object with the HIP module API
(OpenCL and HIP kernels share the runtime's kernel-argument ABI; the hidden NDRange arguments are filled from the
code object's metadata) and launches linearKernel / tileKernel with the reference's 10-argument signature
(resources/kernels/opencl/basic.cl:309-319) over one work block covering the whole image, i.e. what
RendererOpenCL::render does (src/opencl/renderer_opencl.cpp:128-145) when the work block equals the image.

  strict  : -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt (the floating-point model the oracle restates;
            the device library's builtins still use v_rsq_f32 / v_sqrt_f32 / its own sinf, cosf -- see DESIGN.md)
  default : NULL build options, as the reference passes (renderer_opencl.cpp:50)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

KERNELS = ["basic", "basic_lighting", "accumulator", "global_illumination", "global_illumination25", "custom_opencl"]


def available(kernel="basic", flavor="strict"):
    return os.path.exists(os.path.join(REF_DIR, "%s.%s.co" % (kernel, flavor)))


_hip = None
_modules = {}


def _lib():
    global _hip
    if _hip is None:
        import torch  # noqa: F401  (one HIP runtime per process: bind to the copy torch loaded)
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                _hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise ImportError("libamdhip64 not found")
        _hip.hipModuleLoad.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p]
        _hip.hipModuleGetFunction.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_char_p]
        _hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 6 + [ctypes.c_uint, ctypes.c_void_p,
                                                                                         ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        _hip.hipGetErrorString.restype = ctypes.c_char_p
        _hip.hipGetErrorString.argtypes = [ctypes.c_int]
    return _hip


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s: %s" % (what, _lib().hipGetErrorString(rc).decode()))


def _function(kernel, flavor, entry):
    key = (kernel, flavor, entry)
    if key not in _modules:
        L = _lib()
        mod = ctypes.c_void_p()
        _check(L.hipModuleLoad(ctypes.byref(mod), os.path.join(REF_DIR, "%s.%s.co" % (kernel, flavor)).encode()), "hipModuleLoad")
        fn = ctypes.c_void_p()
        _check(L.hipModuleGetFunction(ctypes.byref(fn), mod, entry.encode()), "hipModuleGetFunction")
        _modules[key] = (mod, fn)
    return _modules[key][1]


def _local_size(n, cap):
    for s in range(min(cap, n), 0, -1):
        if n % s == 0:
            return s
    return 1


def render(scene, camera28, W, H, kernel="basic", flavor="strict", mode=0, depth=3, local=None):
    """Returns float32 [H, W, depth] computed by the reference kernel on cuda:0."""
    import torch
    L = _lib()
    dev = torch.device("cuda", 0)
    fn = _function(kernel, flavor, "linearKernel" if mode == 0 else "tileKernel")
    bufs = [torch.from_numpy(np.ascontiguousarray(a).copy()).to(dev) for a in (scene.nodes, scene.prims, scene.materials, scene.lights)]
    cam = torch.from_numpy(np.frombuffer(bytes(camera28), dtype=np.uint8).copy()).to(dev)
    out = torch.zeros((H, W, depth), dtype=torch.float32, device=dev)
    # global size must equal the image (the kernels divide width by get_global_size, basic.cl:321-322)
    lx, ly = local if local else (_local_size(W, 16), _local_size(H, 16))
    if W % lx or H % ly:
        raise ValueError("local size must divide the image")
    ptrs = [ctypes.c_void_p(t.data_ptr()) for t in bufs + [cam, out]]
    scalars = [ctypes.c_uint(0), ctypes.c_uint(W), ctypes.c_uint(H), ctypes.c_uint(depth)]
    args = ptrs + scalars
    argv = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    torch.cuda.synchronize()
    _check(L.hipModuleLaunchKernel(fn, W // lx, H // ly, 1, lx, ly, 1, 0, None, argv, None), "hipModuleLaunchKernel")
    torch.cuda.synchronize()
    return out.cpu().numpy()
