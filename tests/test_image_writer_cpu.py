"""CPU tests of the host library's JPEG writer (lens_trace_amd/host/image_writer.cpp: the reference's ImageWriter,
src/image_writer.cpp:7-24, with a from-scratch baseline encoder instead of stb_image_write).  The files are decoded
with Pillow and compared with the pixels that went in."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lens_trace_amd", "lib", "liblenstrace.so")
Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("host library not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = ctypes.CDLL(LIB)
    L.lt_host_write_jpeg.restype = ctypes.c_int
    L.lt_host_write_jpeg.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
    return L


def write(lib, path, img, quality=100):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    comps = 1 if img.ndim == 2 else img.shape[2]
    rc = lib.lt_host_write_jpeg(str(path).encode(), img.ctypes.data, img.shape[1], img.shape[0], comps, quality)
    assert rc == 0
    return np.asarray(Image.open(path).convert("L" if comps == 1 else "RGB")).astype(int)


def smooth(W, H):
    y, x = np.mgrid[0:H, 0:W]
    return np.stack([127 + 120 * np.sin(x / 17.0), 127 + 120 * np.cos(y / 11.0), (x * 3 + y * 5) % 256], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("W,H", [(64, 64), (100, 100), (37, 21), (8, 8), (1, 1), (257, 19)])
def test_quality_100_round_trip_is_within_two_levels(lib, tmp_path, W, H):
    img = smooth(W, H)
    got = write(lib, tmp_path / "a.jpg", img)
    assert got.shape == img.shape
    assert np.abs(got - img.astype(int)).max() <= 3      # colour-space rounding at all-ones quantisation
    assert np.abs(got - img.astype(int)).mean() < 1.0


def test_noise_flat_and_extreme_images(lib, tmp_path):
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, size=(48, 80, 3), dtype=np.uint8)     # every Huffman symbol class, long codes
    got = write(lib, tmp_path / "n.jpg", noise)
    assert np.abs(got - noise.astype(int)).mean() < 1.5
    for value in (0, 255, 128):
        flat = np.full((33, 47, 3), value, dtype=np.uint8)              # one symbol per table: one-bit codes
        assert np.abs(write(lib, tmp_path / "f.jpg", flat) - value).max() <= 1
    checker = (np.indices((64, 64)).sum(axis=0) % 2 * 255).astype(np.uint8)
    assert np.abs(write(lib, tmp_path / "c.jpg", np.stack([checker] * 3, axis=-1)) - checker[..., None]).max() <= 4   # largest AC amplitudes
    grey = smooth(40, 24)[..., 0]
    assert np.abs(write(lib, tmp_path / "g.jpg", grey) - grey.astype(int)).max() <= 2
    rgba = np.concatenate([smooth(40, 24), np.full((24, 40, 1), 7, dtype=np.uint8)], axis=-1)
    assert np.abs(write(lib, tmp_path / "r.jpg", rgba) - rgba[..., :3].astype(int)).max() <= 3    # a fourth channel is ignored


def test_lower_quality_still_decodes_and_is_smaller(lib, tmp_path):
    img = smooth(128, 96)
    write(lib, tmp_path / "q100.jpg", img, 100)
    got = write(lib, tmp_path / "q50.jpg", img, 50)
    assert np.abs(got - img.astype(int)).mean() < 12.0
    assert os.path.getsize(tmp_path / "q50.jpg") < os.path.getsize(tmp_path / "q100.jpg")
    assert lib.lt_host_write_jpeg(str(tmp_path / "bad.jpg").encode(), None, 4, 4, 3, 100) != 0
