"""The command-line driver (lens_trace_amd/host/lenstrace_cli.cpp -> lib/LensTraceHIP): scene-file schema and defaults of the
reference's SceneParser on CPU (--dry-run), a full render on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.test_host_scene_cpu import WALL_MTL, WALL_OBJ

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "lens_trace_amd", "lib", "LensTraceHIP")


def write_scene(tmp_path, **overrides):
    (tmp_path / "green_wall.obj").write_text(WALL_OBJ)
    (tmp_path / "green_wall.mtl").write_text(WALL_MTL)
    scene = {
        "renderer": {"render_platform": "RENDER_PLATFORM_HIP", "kernel_file_path": "resources/kernels/opencl/basic.cl",
                     "kernel_mode": "KERNEL_MODE_LINEAR", "thread_organization_mode": "THREAD_ORGANIZATION_MODE_MAX_FIT",
                     "image_dimensions": [100, 100, 3]},
        "camera": {"position": [0, 2.5, -50], "pitch": 0, "yaw": 0, "roll": 0},
        "world": {"wall": {"file_path": str(tmp_path / "green_wall.obj")}, "ignored_second_model": {"file_path": "nope.obj"}},
        "output": {"file_path": str(tmp_path / "out.pfm")},
    }
    for k, v in overrides.items():
        scene[k] = v
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(scene, indent=2))
    return p


def test_dry_run_parses_reference_schema_and_defaults(tmp_path):
    p = write_scene(tmp_path)
    out = subprocess.run([CLI, str(p), "--dry-run"], capture_output=True, text=True, check=True).stdout
    assert "platform=RENDER_PLATFORM_HIP" in out and "image=100x100x3" in out and "camera=(0 2.5 -50)" in out
    assert "green_wall.obj" in out and "nope.obj" not in out          # first model only (scene_parser.cpp:106)
    # defaults of the reference's SceneParser when keys are missing
    q = tmp_path / "minimal.json"
    q.write_text('{"world": {"m": {"file_path": "x/y.obj"}}}')
    out = subprocess.run([CLI, str(q), "--dry-run"], capture_output=True, text=True, check=True).stdout
    assert "kernel=resources/kernels/opencl/basic.cl" in out and "image=2048x2048x3" in out and "output=output.jpg" in out
    bad = tmp_path / "bad.json"
    bad.write_text('{"renderer": {"render_platform": "RENDER_PLATFORM_OPTIX"}, "world": {"m": {"file_path": "x/y.obj"}}}')
    assert subprocess.run([CLI, str(bad), "--dry-run"], capture_output=True, text=True).returncode == 1
    bad.write_text('{"renderer": [1, 2,')
    assert subprocess.run([CLI, str(bad), "--dry-run"], capture_output=True, text=True).returncode == 1


@pytest.mark.gpu
def test_cli_renders_the_green_wall(tmp_path):
    p = write_scene(tmp_path)
    subprocess.run([CLI, str(p)], check=True)
    blob = (tmp_path / "out.pfm").read_bytes()
    header, data = blob.split(b"-1.0\n", 1)
    assert header.startswith(b"PF\n100 100\n")
    img = np.frombuffer(data, dtype="<f4").reshape(100, 100, 3)
    assert np.array_equal(img.reshape(-1, 3), np.tile(np.float32([0, 1, 0]), (10000, 1)))   # CorrectColor, through the CLI
    # progressive extension + 8-bit output
    p = write_scene(tmp_path, output={"file_path": str(tmp_path / "out.jpg")}, hip={"frame_first": 1, "frame_count": 4, "accumulate": True})
    subprocess.run([CLI, str(p)], check=True)
    from PIL import Image
    jpg = np.asarray(Image.open(tmp_path / "out.jpg").convert("RGB")).astype(int)
    assert jpg.shape == (100, 100, 3)
    assert np.abs(jpg - np.array([0, 255, 0])).max() <= 2          # quality 100, like the reference's writer
