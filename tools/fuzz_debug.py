#!/usr/bin/env python3
"""GPU box: one fuzz seed of tests/test_gpu_random.py, the worst compositor sum looked at closely.  usage: fuzz_debug.py SEED"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402  (package path)
import helpers as hp  # noqa: E402
import test_gpu_random as tr  # noqa: E402
from oracle import oracle as orc  # noqa: E402

seed = int(sys.argv[1])
P, W, H, deg, cam, sc, bg, bits = tr._fuzz_case(seed)
(ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg, record_bits=bits)
out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw, record_bits=bits)
dLc, dLd = hp.seeds(cam, seed=seed)
gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
em = orc.error_model(st, dLc, dLd, hp.BORDER_REL, hp.BORDER_REL_T)
g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
got = {nm: x for nm, x in zip(hp.GRAD_NAMES, g) if x is not None}
have = hp.compositor_sums(got, P)
want = hp.compositor_sums(gref, P)
mass = em["term_mass"].astype(np.float64)
rel = np.abs(have - want) / (mass + 1e-300)
print("P", P, "W", W, "H", H, "deg", deg, "bits", bits)
for k in range(10):
    i = int(np.argmax(rel[:, k]))
    print("comp %d worst rel-to-mass %.2e at Gaussian %d: got %.6e want %.6e mass %.3e" % (k, rel[i, k], i, have[i, k], want[i, k], mass[i, k]))
i = int(np.argmax(rel[:, 2]))
print("Gaussian", i, "mean2D", st["means2D"][i], "conic_opacity", st["conic_opacity"][i], "radius", ref["radii"][i], "depth", st["depths"][i] if "depths" in st else None)
print("all comps of it: got", have[i], "\nwant", want[i], "\nrel", rel[i])
# which Gaussians are off in comp 2 by > 1e-4 of their mass, and where do they sit
bad = np.nonzero(rel[:, 2] > 1e-4)[0]
print(len(bad), "Gaussians with comp 2 off by > 1e-4 of the mass; their mean x:", st["means2D"][bad][:, 0][:20], "mean y:", st["means2D"][bad][:, 1][:20])
print("conic a of them:", st["conic_opacity"][bad][:, 0][:20], "radii", ref["radii"][bad][:20])
