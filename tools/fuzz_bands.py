#!/usr/bin/env python3
"""GPU box: random scenes (the fuzz cases of tests/test_gpu_random.py) rendered band by band over a RANDOM partition of the tile rows
(tests/test_gpu_tile_band.py::_check_bands: pixels inside a band bit-identical to the whole frame, background outside, radii / counts
add up exactly, every gradient adds up to the whole frame's to fp32 rounding).  Run over 6000..6300: every bitwise / integer check
holds in 300 of 300 scenes; in 4 the summed shares of dL/dcov3D, dL/dscale or dL/drot miss the fixed scenes' 1e-5 of the tensor
maximum by up to 2.7x (ill-conditioned rows of the chain, applied once per share).  usage: fuzz_bands.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
import test_gpu_random as tr  # noqa: E402
import test_gpu_tile_band as tb  # noqa: E402

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    P, W, H, deg, cam, sc, bg3, bits = tr._fuzz_case(seed)
    rng = np.random.default_rng(seed + 3)
    rows = (H + 15) // 16
    cuts = sorted(set(int(c) for c in rng.integers(0, rows + 1, int(rng.integers(1, 5)))) | {0, rows})
    bands = [(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    try:
        tb._check_bands(cam, sc, deg, bands, record_bits=bits, sync=bool(rng.integers(0, 2)), tag="fuzz_band/%d" % seed)
    except AssertionError as e:
        bad += 1
        print(seed, P, W, H, bands, str(e).replace("\n", " ")[:240])
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]))
