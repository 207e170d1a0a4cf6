"""GPU tests (-m gpu) of run-time compiled user programs: a kernelFilePath that is not a built-in name but a
.hip file is compiled with hipRTC at first use and cached by path, like the reference's programMap
(src/opencl/renderer_opencl.cpp:35-54, :67-70)."""
import os

import numpy as np
import pytest

from lens_trace_amd import _capi as C
from lens_trace_amd import scene as sc
from lens_trace_amd.renderer import RendererHIP
from oracle import pyoracle as po
from tests.conftest import GOLDEN
from tests.conftest import oracle_props as RenderPropertiesHIP   # the flavour the CPU oracle reproduces

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "user_kernels")
CAM = sc.camera_bytes(0.0, 2.5, -50.0, 0.0)


@pytest.fixture(scope="module")
def renderer():
    r = RendererHIP(0)
    yield r
    r.close()


def test_user_barycentric_program_equals_builtin_and_oracle(renderer):
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_lens_O0.ltsb")).validate()
    W, H = 128, 128
    user = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(os.path.join(HERE, "barycentric.hip"), (W, H, 3), user, s, pCamera=CAM))
    builtin = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP("examples/custom_kernel/resources/kernels/custom_opencl.cl", (W, H, 3), builtin, s, pCamera=CAM))
    assert np.array_equal(user, builtin)
    assert np.array_equal(user, po.render(s, CAM, W, H, po.CUSTOM))
    # cached by path: the second resolve returns the same program id without recompiling
    a = renderer.resolve_program(os.path.join(HERE, "barycentric.hip"))
    assert a >= 1000 and a == renderer.resolve_program(os.path.join(HERE, "barycentric.hip"))


def test_user_program_with_its_own_rays_matches_oracle_traces(renderer):
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    W = H = 48
    got = np.empty((H, W, 3), dtype=np.float32)
    renderer.render(RenderPropertiesHIP(os.path.join(HERE, "hit_info.hip"), (W, H, 3), got, s, pCamera=CAM))
    # the camera ray of linearKernel (accumulator.cl:304-312), yaw = 0, in float32
    f32 = np.float32
    for y in range(0, H, 3):
        for x in range(0, W, 3):
            fx, fy = f32(x) / f32(W) - f32(0.5), f32(y) / f32(H) - f32(0.5)
            origin = [f32(0.0) + fx, f32(2.5) + fy, f32(-50.0) + f32(0.0), 2.0]
            direction = [f32(0.0) - fx, f32(0.0) - fy, f32(5.0), 0.0]
            hit, prim, tuv = po.trace(s, origin, direction, program=po.ACCUMULATOR)
            if not hit:
                assert got[y, x].tolist() == [-1.0, -1.0, 0.0]
                continue
            assert got[y, x, 0] == tuv[0] and got[y, x, 1] == prim
            pv = s.prim_view[prim]
            b = [f32(f32(np.float64(1.0) - np.float64(tuv[1])) if False else np.float32((1.0 - float(tuv[1])) - float(tuv[2]))), tuv[1], tuv[2]]
            p = [(pv["positionA"][k] * b[0] + pv["positionB"][k] * b[1]) + pv["positionC"][k] * b[2] for k in range(3)]
            hit2, _, _ = po.trace(s, [p[0], p[1], p[2], 1.0], [-direction[0], -direction[1], -direction[2], -0.0], program=po.ACCUMULATOR, ignore=prim)
            assert got[y, x, 2] == 1.0 + 2.0 * hit2


def test_user_program_compile_errors_are_reported(renderer):
    s = sc.load_ltsb(os.path.join(GOLDEN, "green_wall_O0.ltsb"))
    out = np.zeros((8, 8, 3), dtype=np.float32)
    with pytest.raises(C.LensTraceError) as e:
        renderer.render(RenderPropertiesHIP(os.path.join(HERE, "broken.hip"), (8, 8, 3), out, s, pCamera=CAM))
    assert "this_function_does_not_exist" in str(e.value)
    with pytest.raises(C.LensTraceError):
        renderer.render(RenderPropertiesHIP(os.path.join(HERE, "missing.hip"), (8, 8, 3), out, s, pCamera=CAM))


def test_fmod_by_pi_in_closed_form_is_the_librarys_fmod(renderer):
    """random() (acc.cl:63-66) takes fmod(x, pi) in double precision; the kernels compute it as x - n pi with n from one
    multiplication and the remainder from a fused multiply-add (lt_device.hpp: fmod_pi -- fmod is exact, so any way of finding n
    gives the library's bits).  A probe program compares the two, bit for bit, on the arguments random() makes for every pixel of
    a 4K-sized film grid and 480 seeds, their negatives and rescalings, and edge values (multiples of pi, the neighbours of the
    double pi, denormals, 2^40 where the library's own takes over, infinities, NaN)."""
    s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb")).validate()
    W, H = 480, 270
    for frame in range(5):
        out = np.empty((H, W, 3), dtype=np.float32)
        cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, frame)
        renderer.render(RenderPropertiesHIP(os.path.join(HERE, "fmod_check.hip"), (W, H, 3), out, s, pCamera=cam, portableMath=False))
        assert out[..., 1].min() == out[..., 1].max() == 96 * 5 + 2 * 25
        assert out[..., 0].max() == 0.0, "%d pixels saw a difference" % int((out[..., 0] != 0).sum())
