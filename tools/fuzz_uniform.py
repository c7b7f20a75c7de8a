#!/usr/bin/env python3
"""GPU box: fuzz scenes of tests/test_gpu_random.py with ONE colour for all Gaussians (c = accum_rec behind every entry: dL/dalpha is
the background term alone and the colour terms cancel) through the three-layer check.  usage: fuzz_uniform.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
import helpers as hp  # noqa: E402
import test_gpu_random as tr  # noqa: E402

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    P, W, H, deg, cam, sc, bg, bits = tr._fuzz_case(seed)
    sc["shs"][:] = 0.0
    sc["shs"][:, 0, :] = np.array([0.7, -0.2, 0.4], np.float32)  # one colour (SH dc), no view dependence
    try:
        (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg, record_bits=bits)
        out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw, record_bits=bits)
        dLc, dLd = hp.seeds(cam, seed=seed)
        hp.check_backward(cam, deg, out, args, st, dLc, dLd, "uniform/%d" % seed)
    except AssertionError as e:
        bad += 1
        print(seed, str(e)[:300])
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]))
