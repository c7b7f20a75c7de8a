#!/usr/bin/env python3
"""Randomized parity sweep: N random scenes / cameras / options, HIP path vs the CPU oracle, same checks as
tests/test_gpu_random.py::test_random_scene_parity (integers exact, images with a threshold-flip budget, gradients 3e-3).
usage: fuzz_parity.py [N=40] [seed0=1000]      (GPU box; prints one line per case, exits non-zero on the first failure)"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gs-slam-analytica_jacobian_amd")):
    sys.path.insert(0, p)
import helpers as hp  # noqa: E402
from gsaj import rasterizer as C, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def one(seed):
    rng = np.random.default_rng(seed)
    P = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 3000, 8000]))
    W, H = int(rng.integers(16, 400)), int(rng.integers(16, 300))
    coeffs = int(rng.choice([1, 4, 9, 16]))
    deg = int(rng.integers(0, int(round(math.sqrt(coeffs)))))
    zlo = float(rng.uniform(0.15, 1.5))
    zr = (zlo, zlo + float(rng.uniform(0.2, 6.0)))
    lo = float(rng.uniform(math.log(0.002), math.log(0.05)))
    ls = (lo, lo + float(rng.uniform(0.1, 3.0)))
    olo = float(rng.uniform(0.003, 0.6))
    orng = (olo, min(1.0, olo + float(rng.uniform(0.05, 0.6))))
    cam = hp.small_camera(W, H, f=float(rng.uniform(0.5, 1.5)) * W, orthonormal=bool(rng.integers(0, 2)))
    sc = syn.make_scene(P, seed, cam, z_range=zr, log_scale_range=ls, opacity_range=orng, sh_coeffs=coeffs, margin=float(rng.uniform(0.0, 0.4)))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"], ("R", R, ref["num_rendered"])
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    assert (dbg["n_contrib"].astype(np.int64) != st["n_contrib"].astype(np.int64)).mean() <= 5e-4
    hp.assert_image_close(color.cpu().numpy(), ref["color"], 2e-4)
    hp.assert_image_close(depth.cpu().numpy(), ref["depth"], 2e-4)
    dLc, dLd = hp.seeds(cam, seed=seed)
    gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
    g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dtau"]
    worst = 0.0
    for nm, got in zip(names, g[:9]):
        want = gref[nm]
        if np.abs(want).max() == 0:
            assert float(got.abs().max()) == 0, nm
            continue
        e = hp.rel_err(got.cpu().numpy().reshape(want.shape), want)
        worst = max(worst, e)
        assert e < 3e-3, (nm, e)
    e = hp.rel_err(g[9].cpu().numpy(), gref["dL_dtau_sum"]) if np.abs(gref["dL_dtau_sum"]).max() > 0 else 0.0
    assert e < 3e-3, ("tau_sum", e)
    return "seed %d P=%d %dx%d deg=%d/%d R=%d I=%d worst grad err %.1e" % (seed, P, W, H, deg, coeffs, R, st["interactions"], max(worst, e))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    for s in range(s0, s0 + n):
        try:
            print(one(s), flush=True)
        except Exception as ex:  # noqa: BLE001
            print("FAILED seed %d: %r" % (s, ex), flush=True)
            raise


if __name__ == "__main__":
    main()
