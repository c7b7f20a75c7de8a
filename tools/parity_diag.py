#!/usr/bin/env python3
"""GPU box: where do the HIP-vs-oracle gradient differences come from?  For a few scenes prints, per compositor sum,
the worst |err| / sum|terms| and |err| / tensor max, the Gaussian it belongs to, and the chain errors on the device's sums."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gs-slam-analytica_jacobian_amd")):
    sys.path.insert(0, p)
import helpers as hp  # noqa: E402
from gsaj import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random as tr  # noqa: E402

orc.set_threads(16)


def report(tag, cam, sc, deg, bg=(0, 0, 0), bits=32, seed=1):
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg, record_bits=bits)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw, record_bits=bits)
    dLc, dLd = hp.seeds(cam, seed=seed)
    gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
    em = orc.error_model(st, dLc, dLd)
    g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
    got = dict(zip(hp.GRAD_NAMES, g))
    P = sc["means3D"].shape[0]
    have, want, mass = hp.compositor_sums(got, P), hp.compositor_sums(gref, P), em["term_mass"].astype(np.float64)
    cond, flip = em["cond_slack"].astype(np.float64), em["flip_budget"].astype(np.float64)
    err = np.abs(have - want)
    for K in (1.0, 2.0, 4.0):
        b0 = hp.MASS_TOL * mass + K * cond + 1e-9 * np.abs(want).max(axis=0, keepdims=True) + 1e-37
        print("  COND_K=%.0f: max err/(mass+cond bound) %.2f over %d elements beyond it; with flip budget x1.5: max %.3f; x1.0: %.3f; border px %d"
              % (K, (err / b0).max(), (err > b0).sum(), (err / (b0 + 1.5 * flip)).max(), (err / (b0 + flip)).max(), em["border_mask"].sum()))
    gmax = np.abs(want).max(axis=0)
    print("== %s P=%d R=%d I=%d" % (tag, P, ref["num_rendered"], st["interactions"]))
    for c in range(10):
        r = err[:, c] / (mass[:, c] + 1e-30)
        r[mass[:, c] == 0] = 0
        i = int(np.argmax(r))
        j = int(np.argmax(err[:, c]))
        print("  sum%d: worst err/mass %.2e (G%d: got %.4e want %.4e mass %.3e tiles %d radius %d)  worst err/max %.2e (G%d err/mass %.2e)  p99.9 err/mass %.2e"
              % (c, r[i], i, have[i, c], want[i, c], mass[i, c], st["tiles_touched"][i], st["radii"][i], err[j, c] / (gmax[c] + 1e-30), j,
                 err[j, c] / (mass[j, c] + 1e-30), np.quantile(r, 0.999)))
    ch = orc.chain(st, hp._np(got["dL_dmean2D"]), hp._np(got["dL_dconic"]), hp._np(got["dL_dcolor"]), hp._np(got["dL_ddepth"]), cam["projmatrix_raw"])
    for nm in hp.CHAIN_NAMES:
        if ch[nm].size and np.abs(ch[nm]).max() > 0:
            print("  chain %-10s row(1e-3 floor) %.2e  row(1e-2 floor) %.2e  global %.2e | vs full oracle global %.2e"
                  % (nm, hp.row_rel_err(hp._np(got[nm]), ch[nm], floor=1e-3), hp.row_rel_err(hp._np(got[nm]), ch[nm], floor=1e-2),
                     hp.rel_err(hp._np(got[nm]).reshape(ch[nm].shape), ch[nm]), hp.rel_err(hp._np(got[nm]).reshape(ch[nm].shape), gref[nm])))
    print("  tau_sum vs chain-of-device-sums %.2e   vs full oracle %.2e" % (hp.rel_err(hp._np(got["dL_dtau_sum"]), ch["dL_dtau_sum"]),
                                                                             hp.rel_err(hp._np(got["dL_dtau_sum"]), gref["dL_dtau_sum"])))
    # images
    col = out[1].cpu().numpy()
    e = np.abs(col - ref["color"]) / np.abs(ref["color"]).max()
    bad = (e > hp.IMG_TOL).any(axis=0)
    nb = 0
    for py, px in zip(*np.nonzero(bad)):
        nb += not hp.borderline_pixel(st, int(px), int(py))
    ok = e[:, ~bad]
    print("  image: max err %.2e, %d pixels beyond IMG_TOL, %d of them NOT borderline; max err of in-tolerance pixels %.2e" % (e.max(), bad.sum(), nb, ok.max() if ok.size else 0))
    sys.stdout.flush()


for wl in ("cfg2", "cfg3", "cfg5"):
    cam, sc = syn.config_scene(wl)
    M = sc["shs"].shape[1]
    report(wl, cam, sc, int(round(M ** 0.5)) - 1, seed=31)
for seed in (5052, 5059, 5004):
    P, W, H, deg, cam, sc, bg, bits = tr._fuzz_case(seed)
    report("fuzz%d %dx%d bits%d" % (seed, W, H, bits), cam, sc, deg, bg=bg, bits=bits, seed=seed)
c = tr.CASES[4]
P, W, H, seed, deg, coeffs, zr, ls, orng, ortho = c
cam = hp.small_camera(W, H, f=0.8 * W, orthonormal=ortho)
sc = syn.make_scene(P, seed, cam, z_range=zr, log_scale_range=ls, opacity_range=orng, sh_coeffs=coeffs, margin=0.2)
report("random P1000", cam, sc, deg, bg=(0.3, 0.1, 0.7), seed=seed)
