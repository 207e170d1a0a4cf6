#!/usr/bin/env python3
"""Diagnostic: compare the images two ab_dump.py runs left under gpurun_out/ and delete them."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a, b = sys.argv[1], sys.argv[2]
for fa in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "ab_%s_*.npy" % a))):
    fb = fa.replace("ab_%s_" % a, "ab_%s_" % b)
    x, y = np.load(fa), np.load(fb)
    diff = np.any(x != y, axis=2)
    print(os.path.basename(fa), "identical" if not diff.any() else "DIFFERENT in %d pixels, rms %.3g" % (
        diff.sum(), np.sqrt(np.mean((x.astype(np.float64) - y) ** 2))))
    if len(sys.argv) < 4:   # a third argument keeps the files (more comparisons against the same base follow)
        os.remove(fa)
        os.remove(fb)
