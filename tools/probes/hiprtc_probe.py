"""Probe: does hipRTC compile for gfx950 on the box, in a process that already holds torch's HIP runtime?"""
import ctypes
import torch  # noqa: F401
torch.cuda.init()
for name in ("libhiprtc.so.7", "libhiprtc.so"):
    try:
        L = ctypes.CDLL(name)
        print("loaded", name)
        break
    except OSError as e:
        print("cannot load", name, e)
src = b'#include <hip/hip_runtime.h>\nextern "C" __global__ void k(float* o) { o[threadIdx.x] = __builtin_fmaf(2.0f, threadIdx.x, 1.0f); }\n'
prog = ctypes.c_void_p()
rc = L.hiprtcCreateProgram(ctypes.byref(prog), src, b"k.hip", 0, None, None)
opts = (ctypes.c_char_p * 3)(b"--offload-arch=gfx950", b"-O3", b"-ffp-contract=off")
rc2 = L.hiprtcCompileProgram(prog, 3, opts)
n = ctypes.c_size_t()
L.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
log = ctypes.create_string_buffer(n.value + 1)
L.hiprtcGetProgramLog(prog, log)
L.hiprtcGetCodeSize(prog, ctypes.byref(n))
print("create", rc, "compile", rc2, "code bytes", n.value, "log:", log.value[:300])
