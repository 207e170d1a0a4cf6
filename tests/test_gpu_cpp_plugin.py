"""GPU test (-m gpu): the reference's own renderer tests (tests/opencl_renderer_test.cc:5-228 -- ValidEngine, ValidBuffer,
CustomBlockSize, KernelMode, CorrectColor), re-stated in C++ against `class RendererHIP : public Renderer`
(tests/cpp/hip_renderer_test.cpp, built by lens_trace_amd/host/Makefile into lib/hip_renderer_test), plus the two extension
structs.  This is the C++ plugin surface a user of the reference switches to; everything else in the suite drives the same C ABI
through ctypes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lens_trace_amd", "lib", "hip_renderer_test")


def test_restated_reference_renderer_tests_pass(tmp_path):
    assert os.path.exists(BIN), "lib/hip_renderer_test is not built (run __graft_entry__.build())"
    p = subprocess.run([BIN, str(tmp_path)], capture_output=True, text=True, timeout=300)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 failed, 0 of 7 tests failed" in p.stdout
    for name in ("CreateEngineTEST.ValidEngine", "RenderBufferTEST.ValidBuffer", "RenderBufferTEST.CustomBlockSize",
                 "RenderBufferTEST.KernelMode", "RenderBufferTEST.CorrectColor", "RenderBufferTEST.ProgressiveExtension",
                 "RenderBufferTEST.BackendExtension"):
        assert "[       OK ] " + name in p.stdout
