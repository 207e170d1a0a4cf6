#!/usr/bin/env python3
"""Generates tests/golden/ from the REFERENCE (this container only; /root/reference is read, never copied).

 1. scene buffers: oracle/_ref/ref_host_dump_{O0,O2} (the reference's own Model /
    AccelerationStructureExplicit / Camera classes, built by oracle/build_ref.sh) dump the five
    raw upload buffers for the three .obj models the reference ships.  -O0 is how the reference's CI
    builds (no CMAKE_BUILD_TYPE).  (The builder's uninitialised centroid bounds, SURVEY Q1, make the
    BVH depend on stack garbage; in this container -O0 and -O2 gave identical buffers, so only -O0 is
    kept.  The multi-primitive-leaf quirk Q2 is exercised by a hand-made scene in tests/.)
 2. expected pixels: the CPU oracle's output per program on those buffers (small images), so GPU
    parity tests can also compare against committed data.  For `basic` on green_wall this equals the
    reference's own known answer (tests/opencl_renderer_test.cc:185-228: every pixel (0,1,0)).

Fixtures are data only (inputs + expected outputs)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def dump(opt, model, name):
    out = os.path.join(GOLD, name + ".ltsb")
    if os.path.exists(out):      # frozen: the reference's BVH builder reads uninitialised memory and changes run to run
        return out
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_host_dump_" + opt), "resources/models/" + model, out],
                   cwd=REF, check=True)
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    dump("O0", "green_wall.obj", "green_wall_O0")
    dump("O0", "cornell_box.obj", "cornell_box_O0")
    dump("O0", "cornell_box_lens.obj", "cornell_box_lens_O0")
    cases = [  # (scene, program, mode, W, H, frameCount, yaw)
        ("green_wall_O0", "basic", 0, 100, 100, 0, 0.0),
        ("cornell_box_O0", "basic", 0, 128, 128, 0, 0.0),
        ("cornell_box_O0", "basic", 0, 96, 64, 0, 0.03),
        ("cornell_box_lens_O0", "basic", 0, 128, 128, 0, 0.0),
        ("cornell_box_O0", "accumulator", 0, 128, 128, 0, 0.0),
        ("cornell_box_O0", "accumulator", 0, 128, 128, 1, 0.0),
        ("cornell_box_O0", "accumulator", 1, 128, 128, 7, 0.0),
        ("cornell_box_O0", "accumulator", 0, 96, 64, 3, -0.02),
        ("cornell_box_O0", "global_illumination", 0, 128, 128, 0, 0.0),
        ("cornell_box_O0", "global_illumination", 1, 128, 128, 1, 0.0),
        ("cornell_box_O0", "basic_lighting", 0, 64, 64, 0, 0.0),
        ("cornell_box_O0", "global_illumination25", 0, 64, 64, 2, 0.0),
        ("cornell_box_lens_O0", "custom_opencl", 0, 128, 128, 0, 0.0),
    ]
    index = []
    for scene_name, prog, mode, W, H, fc, yaw in cases:
        s = sc.load_ltsb(os.path.join(GOLD, scene_name + ".ltsb"))
        cam = sc.camera_bytes(0.0, 2.5, -50.0, yaw, 0.0, 0.0, fc)
        img, st = po.render(s, cam, W, H, po.PROGRAMS[prog], mode, threads=8, want_stats=True)
        tag = "%s__%s_m%d_%dx%d_f%d_y%s" % (scene_name, prog, mode, W, H, fc, ("%g" % yaw).replace("-", "n").replace(".", "p"))
        np.save(os.path.join(GOLD, tag + ".npy"), img)
        index.append("%s %s %s %d %d %d %d %r %d %d %d %d" % (tag, scene_name, prog, mode, W, H, fc, yaw, st["rays"],
                                                            st["shadow_rays"], st["node_visits"], st["tri_tests"]))
        print(index[-1])
    with open(os.path.join(GOLD, "index.txt"), "w") as f:
        f.write("# tag scene program mode W H frameCount yaw rays shadow_rays node_visits tri_tests\n")
        f.write("\n".join(index) + "\n")


if __name__ == "__main__":
    main()
