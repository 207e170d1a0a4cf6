"""Compiles a user program with hipRTC exactly as lt_hip_resolve_program does (no GPU needed) and prints the log."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = ctypes.CDLL("libhiprtc.so.7")
user = open(sys.argv[1]).read()
src = ('#define LT_USER_PROGRAM 1\n#include "lt_kernel.hpp"\n#line 1 "%s"\n' % sys.argv[1] + user +
       '\nextern "C" __global__ __launch_bounds__(64) void lt_user_kernel_lds(SceneDev sc, FrameParams fp, float* out, unsigned long long* stats, uint32_t* queues) {\n'
       '  render_kernel_body<kUser, Config<false, false, false>>(sc, fp, out, stats, queues);\n}\n').encode()
prog = ctypes.c_void_p()
L.hiprtcCreateProgram(ctypes.byref(prog), src, b"lt_user_program.hip", 0, None, None)
opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-ffp-contract=off", ("-I" + os.path.join(ROOT, "lens_trace_amd", "csrc")).encode()]
rc = L.hiprtcCompileProgram(prog, len(opts), (ctypes.c_char_p * len(opts))(*opts))
n = ctypes.c_size_t()
L.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
log = ctypes.create_string_buffer(n.value + 1)
L.hiprtcGetProgramLog(prog, log)
print("rc", rc)
print(log.value.decode()[:3000])
