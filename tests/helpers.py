"""Shared helpers of the parity tests: run the CPU oracle and the HIP library on the same
seeded synthetic scene and hand back comparable NumPy views."""
import math

import numpy as np

from gsaj import synthetic as syn
from oracle import oracle as orc


def small_camera(W, H, f=None, noisy=True, orthonormal=False):
    f = f if f is not None else 0.9 * W
    return syn.fixture_camera(noisy=noisy, orthonormal=orthonormal, W=W, H=H, fx=f, fy=f, cx=W / 2 - 0.5, cy=H / 2 - 0.5)


SCENES = {
    # name: (P, W, H, seed, sh_degree, kwargs)
    "n15_640x480": dict(P=15, W=640, H=480, seed=15, deg=3, cam=lambda: syn.fixture_camera(noisy=True),
                        scene=dict(z_range=(0.8, 1.6), log_scale_range=(math.log(0.01), math.log(0.05)))),
    "p2000_160x120": dict(P=2000, W=160, H=120, seed=2, deg=3, cam=lambda: small_camera(160, 120, orthonormal=True),
                          scene=dict(z_range=(1.0, 5.0), log_scale_range=(math.log(0.01), math.log(0.08)))),
    "p500_100x75_sh0": dict(P=500, W=100, H=75, seed=3, deg=0, cam=lambda: small_camera(100, 75),
                            scene=dict(z_range=(0.6, 3.0), log_scale_range=(math.log(0.01), math.log(0.1)), sh_coeffs=1)),
    "p6000_640x480_sh1": dict(P=6000, W=640, H=480, seed=4, deg=1, cam=lambda: syn.fixture_camera(noisy=True, orthonormal=True),
                              scene=dict(z_range=(1.0, 6.0), log_scale_range=(math.log(0.005), math.log(0.05)), sh_coeffs=4)),
    "p300_behind_64x48": dict(P=300, W=64, H=48, seed=5, deg=2, cam=lambda: small_camera(64, 48),
                              scene=dict(z_range=(-1.0, 2.0), log_scale_range=(math.log(0.02), math.log(0.2)), sh_coeffs=9, margin=0.6)),
}


def make(name):
    spec = SCENES[name]
    cam = spec["cam"]()
    sc = syn.make_scene(spec["P"], spec["seed"], cam, **spec["scene"])
    return cam, sc, spec["deg"]


def oracle_forward(cam, sc, deg, bg=(0.0, 0.0, 0.0), precomp=False, record_bits=32):
    kw = dict(sh_degree=deg)
    if precomp:
        rng = np.random.default_rng(99)
        kw.update(colors_precomp=rng.uniform(0, 1, size=(sc["means3D"].shape[0], 3)).astype(np.float32),
                  cov3D_precomp=syn.covariance6(sc["scales"], sc["rotations"]))
    else:
        kw.update(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    return orc.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                       cam["tanfovx"], cam["tanfovy"], cam["W"], cam["H"], np.asarray(bg, np.float32),
                       record_bits=record_bits, **kw), kw


def seeds(cam, seed=0):
    rng = np.random.default_rng(seed)
    H, W = cam["H"], cam["W"]
    return (rng.normal(size=(3, H, W)).astype(np.float32) / (3 * H * W),
            rng.normal(size=(1, H, W)).astype(np.float32) / (H * W))


def gpu_forward(cam, sc, deg, bg=(0.0, 0.0, 0.0), kw=None, device="cuda:0", record_bits=32):
    """Through the C ABI (gsaj.rasterizer = the `_C` module of the drop-in package)."""
    import torch
    from gsaj import rasterizer as C

    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    e = torch.empty(0)
    g = lambda k: t(kw[k]) if (kw and k in kw and kw[k] is not None and k != "sh_degree") else e  # noqa: E731
    args = dict(bg=t(np.asarray(bg, np.float32)), means3D=t(sc["means3D"]), colors=g("colors_precomp"),
                opacity=t(sc["opacities"]), scales=g("scales"), rotations=g("rotations"), cov3D=g("cov3D_precomp"),
                view=t(cam["viewmatrix"]), proj=t(cam["projmatrix"]), proj_raw=t(cam["projmatrix_raw"]), sh=g("shs"),
                campos=t(cam["campos"]))
    out = C.rasterize_gaussians(args["bg"], args["means3D"], args["colors"], args["opacity"], args["scales"],
                                args["rotations"], 1.0, args["cov3D"], args["view"], args["proj"], args["proj_raw"],
                                cam["tanfovx"], cam["tanfovy"], cam["H"], cam["W"], args["sh"], deg, args["campos"],
                                False, False, record_bits=record_bits)
    return out, args


def gpu_backward(cam, deg, fwd_out, args, dLc, dLd, device="cuda:0"):
    import torch
    from gsaj import rasterizer as C

    R, color, radii, geom, binning, img, depth, opacity, n_touched = fwd_out
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    return C.rasterize_gaussians_backward(args["bg"], args["means3D"], radii, args["colors"], args["scales"],
                                          args["rotations"], 1.0, args["cov3D"], args["view"], args["proj"],
                                          args["proj_raw"], cam["tanfovx"], cam["tanfovy"], t(dLc), t(dLd), args["sh"],
                                          deg, args["campos"], geom, R, binning, img, False)


# ---- tolerances of the HIP-vs-oracle comparisons (stated once, used by every GPU parity test) -------------------------
# The kernels accumulate in fp32 in a fixed order that differs from the oracle's (fp64 per-Gaussian sums), use
# v_exp_f32 / v_rcp_f32 where the oracle uses libm expf and a true division.  Measured worst cases over every test
# scene are written to gpurun_out/parity_errors.jsonl (GSAJ_ERRLOG=1); the limits below are ~10-30x above them.
IMG_TOL = 5e-5       # images, relative to the image's max, for pixels with no cut-off borderline contributor
GRAD_TOL = 5e-5      # every gradient tensor, max |err| relative to the tensor's max (Gaussians untouched by borderline pixels)
GRAD_TOL_FLIPPED = 1e-2  # same, over all Gaussians incl. those where a cut-off decision may fall either way (sanity net: (A) bounds them exactly)
# reverse-compositor sums, per Gaussian and component: |err| <= MASS_TOL * sum|term| + COND_K * cond_slack + FLIP_K * flip_budget
# (oracle.error_model: a different fp32 summation order; the rounding of power / of the T recovery that ANY fp32 evaluation
# has; pixels where a cut-off may legitimately fall either way, re-evaluated both ways)
MASS_TOL = 2e-5
COND_K = 4.0
FLIP_K = 1.5
CHAIN_ROW_TOL = 1e-4  # per-Gaussian chain on the device's own sums, per row: relative to max(|row|, CHAIN_FLOOR * tensor max)
CHAIN_FLOOR = 1e-3
CHAIN_COND_K = 16.0   # ... or within this many times what a +-1 ulp perturbation of the chain's fp32 inputs does to the fp64 chain (chain_sensitivity)
CHAIN_K = 4.0
ROW_FLOOR = 1e-2
BORDER_REL = 1e-5    # a contributor is "borderline" if alpha is within this (relative) of 1/255 ...
BORDER_REL_T = 1e-4  # ... or T(1-alpha) within this of 1e-4 (T carries the rounding of every nearer contributor)


def _errlog(tag, **kv):
    import json
    import os
    if os.environ.get("GSAJ_ERRLOG"):
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "parity_errors.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(tag=tag, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in kv.items()})) + "\n")


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


def row_rel_err(got, want, floor=ROW_FLOOR):
    """Worst per-row error: every row (one Gaussian) is judged against ITS OWN magnitude, so a wrong small-magnitude
    gradient cannot hide behind the tensor's largest entry; rows smaller than floor * (tensor max) are judged against
    that floor (their value is the difference of larger, individually rounded terms)."""
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64).reshape(want.shape)
    if want.ndim == 1:
        want, got = want[:, None], got[:, None]
    want2, got2 = want.reshape(want.shape[0], -1), got.reshape(want.shape[0], -1)
    gmax = np.abs(want2).max() + 1e-30
    scale = np.maximum(np.abs(want2).max(axis=1), floor * gmax)
    return float((np.abs(got2 - want2).max(axis=1) / scale).max())


GRAD_NAMES = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot",
              "dL_dtau", "dL_dtau_sum", "dL_dconic", "dL_ddepth"]
CHAIN_NAMES = ["dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dtau"]


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def compositor_sums(d, P):
    """The reverse compositor's 10 per-Gaussian sums as one [P,10] array (order of the oracle's term_mass)."""
    m2, cn = _np(d["dL_dmean2D"]).reshape(P, 3), _np(d["dL_dconic"]).reshape(P, 4)
    return np.concatenate([m2[:, :2], cn[:, [0, 1, 3]], _np(d["dL_dopacity"]).reshape(P, 1), _np(d["dL_dcolor"]).reshape(P, 3),
                           _np(d["dL_ddepth"]).reshape(P, 1)], axis=1).astype(np.float64)


def chain_sensitivity(st, sums, projmatrix_raw, trials=4, seed=0):
    """Per output row of the per-Gaussian chain: how far the chain CARRIED IN FP64 moves when every fp32 input it reads
    (means, scales, rotations / covariances, SH coefficients, the compositor sums) is nudged by +-1 ulp, worst of `trials`
    random sign patterns.  That is (a sample of) the row's condition number times eps: no fp32 evaluation, the reference's
    included, can be held to less -- rows where a c - b^2 or the antisymmetric part of dL/dR cancels have it large.
    Also returned: noise32, per row the largest |fp32 chain - fp64 chain| of the ORACLE over the original and the nudged inputs
    (1 + trials samples): cancellations INSIDE the chain (needle-shaped Gaussians: the three terms of d(conic)/d(cov2D))
    amplify the rounding of every fp32 evaluation but are invisible to an input perturbation, which moves the terms together."""
    from oracle import oracle as orc

    rng = np.random.default_rng(seed)
    base = orc.chain(st, *sums, projmatrix_raw, f64=True)

    def nudge(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, np.float32)
        return np.nextafter(a, np.where(rng.integers(0, 2, a.shape) > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))

    sens = {nm: np.zeros(base[nm].shape[0]) for nm in CHAIN_NAMES}
    noise32 = {nm: np.zeros(base[nm].shape[0]) for nm in CHAIN_NAMES}

    def rows(a):
        return np.asarray(a, np.float64).reshape(a.shape[0], -1)

    o32 = orc.chain(st, *sums, projmatrix_raw)
    for nm in CHAIN_NAMES:
        if base[nm].size:
            noise32[nm] = np.abs(rows(o32[nm]) - rows(base[nm])).max(axis=1)
    for _ in range(trials):
        st2 = dict(st)
        st2["inputs"] = dict(st["inputs"])
        for k in ("means3D", "scales", "rotations", "cov3D_precomp", "shs"):
            st2["inputs"][k] = nudge(st["inputs"].get(k))
        if st.get("cov3D") is not None:
            st2["cov3D"] = nudge(st["cov3D"])
        sums2 = [nudge(x) for x in sums]
        out = orc.chain(st2, *sums2, projmatrix_raw, f64=True)
        out32 = orc.chain(st2, *sums2, projmatrix_raw)
        for nm in CHAIN_NAMES:
            if base[nm].size:
                sens[nm] = np.maximum(sens[nm], np.abs(rows(out[nm]) - rows(base[nm])).max(axis=1))
                # the SAME nudged inputs through the oracle's fp32 chain: other input bits, other rounding pattern inside the
                # chain -- the scatter of these samples is the rounding error an fp32 evaluation of this row has
                noise32[nm] = np.maximum(noise32[nm], np.abs(rows(out32[nm]) - rows(out[nm])).max(axis=1))
    return base, sens, noise32, o32


def assert_grads_close(g, gref, tag, st=None, projmatrix_raw=None, tol=GRAD_TOL, skip=()):
    """Every output of rasterize_gaussians_backward (12-tuple, GRAD_NAMES order) against the oracle.  Three layers:
    (A) the reverse compositor's 10 per-Gaussian sums, element by element, against the oracle's fp64-accumulated sums, within
        what two correct fp32 evaluations may differ by (oracle.error_model, gref["error_model"]): MASS_TOL * sum|term| for the
        summation order, COND_K * cond_slack for the rounding of power and of the T recovery, FLIP_K * flip_budget for pixels
        where a cut-off can fall either way.  This judges every Gaussian against ITS OWN terms, however small its net
        gradient is next to the tensor's largest;
    (B) the per-Gaussian chain (mean3D, cov3D, SH, scale, rotation, per-Gaussian tau) evaluated on the DEVICE's compositor
        sums, row by row against the chain carried in fp64: the error must be within CHAIN_ROW_TOL of the row's own magnitude
        (floored at CHAIN_FLOOR of the tensor max) or -- for ill-conditioned rows (a c - b^2 or the antisymmetric part of dL/dR
        cancels), where NO fp32 evaluation reaches that -- within CHAIN_COND_K times what a +-1 ulp perturbation of the chain's
        fp32 inputs does to the fp64 result (chain_sensitivity: the row's conditioning, measured), or within CHAIN_K times the
        error of the oracle's own fp32 chain ON THAT ROW, worst of five samples (original + four nudged inputs; needle-shaped
        Gaussians: the three terms of d(conic)/d(cov2D) cancel inside the chain, which rounds every fp32 evaluation alike but is
        invisible to an input perturbation);
    (C) end to end against the full oracle, max |err| relative to the tensor's max: <= `tol` (GRAD_TOL) over the Gaussians no
        borderline pixel touches; over ALL of them, row by row, within (A)'s bounds on the ten sums carried through the chain (it
        is linear in them: nine fp64 chain evaluations give |J| bound exactly) plus (B)'s allowance -- a flipped contributor
        moves a small Gaussian's gradient by one pixel's worth, and that is now priced per Gaussian instead of by a constant;
        GRAD_TOL_FLIPPED stays as the net for callers without the oracle state and for dL/dtau summed over the Gaussians."""
    from oracle import oracle as orc

    got = {nm: x for nm, x in zip(GRAD_NAMES, g) if x is not None and nm not in skip}
    worst = {}
    chain_allow, chain_err32 = {}, {}  # per chain tensor, per row: what (B) allows / the oracle's own fp32 chain error
    chain_noise = {}  # per chain tensor: how far the oracle's own fp32 chain is from the fp64 chain (tensor-max relative)
    if st is not None and projmatrix_raw is not None and "dL_dmean2D" in got:  # (B)
        sums = (_np(got["dL_dmean2D"]), _np(got["dL_dconic"]), _np(got["dL_dcolor"]), _np(got["dL_ddepth"]))
        truth, sens, noise32, o32 = chain_sensitivity(st, sums, projmatrix_raw)
        for nm in CHAIN_NAMES:
            if nm not in got or truth[nm].size == 0 or np.abs(truth[nm]).max() == 0:
                continue
            t = truth[nm].reshape(truth[nm].shape[0], -1)
            x = _np(got[nm]).astype(np.float64).reshape(t.shape)
            scale = np.maximum(np.abs(t).max(axis=1), CHAIN_FLOOR * np.abs(t).max())
            err = np.abs(x - t).max(axis=1)
            err32 = noise32[nm]  # the oracle's fp32 chain on the SAME row, worst of 1 + 4 samples
            allowed = np.maximum(np.maximum(CHAIN_ROW_TOL * scale, CHAIN_COND_K * sens[nm]), CHAIN_K * err32)
            chain_allow[nm], chain_err32[nm] = allowed, err32
            worst[nm + "/chain_row"] = float((err / scale).max())
            worst[nm + "/chain_row_oracle32"] = row_rel_err(o32[nm], truth[nm], floor=CHAIN_FLOOR)
            worst[nm + "/chain_row_over_allowed"] = float((err / allowed).max())
            worst[nm + "/chain_rows_needing_conditioning"] = int((err > CHAIN_ROW_TOL * scale).sum())
            chain_noise[nm] = rel_err(o32[nm], truth[nm])
            i = int(np.argmax(err / allowed))
            assert err[i] <= allowed[i], (tag, nm, "per-Gaussian chain, row %d: error %.3e; allowed %.3e = max(%.0e x row scale %.3e, %g x "
                                          "sensitivity to +-1 ulp inputs %.3e, %g x the fp32 oracle's worst error on this row over 5 samples %.3e)"
                                          % (i, err[i], allowed[i], CHAIN_ROW_TOL, scale[i], CHAIN_COND_K, sens[nm][i], CHAIN_K, err32[i]))
        if "dL_dtau_sum" in got and np.abs(truth["dL_dtau_sum"]).max() > 0:
            e_hip, e_o32 = rel_err(_np(got["dL_dtau_sum"]), truth["dL_dtau_sum"]), rel_err(o32["dL_dtau_sum"], truth["dL_dtau_sum"])
            worst["dL_dtau_sum/chain"], worst["dL_dtau_sum/chain_oracle32"] = e_hip, e_o32
            assert e_hip < max(1e-5, CHAIN_K * e_o32), (tag, "dL_dtau_sum on the device's sums: device %.2e, fp32 oracle %.2e (vs fp64)" % (e_hip, e_o32))
    em = gref.get("error_model")
    # Gaussians no borderline pixel touches: their sums involve no cut-off decision that could fall either way
    clean = (em["flip_budget"].max(axis=1) == 0) if em is not None else None
    # What layer (A) allows the ten compositor sums to differ by, carried through the per-Gaussian chain -- which is LINEAR in
    # the sums, so chain64(bound_k e_k) is column k of |J| bound exactly: per row of every chain output, the largest end-to-end
    # difference two evaluations inside (A)'s bounds can show.  With it (C) needs no constant for the Gaussians a borderline
    # pixel touches: |device - oracle| <= propagated (A) bound + the chain allowance of (B) + the oracle's own fp32 chain error.
    derived = None
    if em is not None and chain_allow and all(k in got for k in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_ddepth")):
        P = gref["dL_dopacity"].shape[0]
        mass, cond, flip = (em[k].astype(np.float64) for k in ("term_mass", "cond_slack", "flip_budget"))
        want10 = compositor_sums(gref, P)
        bound10 = MASS_TOL * mass + COND_K * cond + FLIP_K * flip + 1e-9 * np.abs(want10).max(axis=0, keepdims=True) + 1e-37
        derived = {nm: 0.0 for nm in chain_allow}
        for k in (0, 1, 2, 3, 4, 6, 7, 8, 9):  # (5 = opacity: not a chain input)
            m2, cn, dc, dd = np.zeros((P, 3)), np.zeros((P, 4)), np.zeros((P, 3)), np.zeros((P, 1))
            if k < 2:
                m2[:, k] = bound10[:, k]
            elif k < 5:
                cn[:, (0, 1, 3)[k - 2]] = bound10[:, k]
            elif k < 9:
                dc[:, k - 6] = bound10[:, k]
            else:
                dd[:, 0] = bound10[:, k]
            col = orc.chain(st, m2, cn, dc, dd, projmatrix_raw, f64=True)
            for nm in chain_allow:
                derived[nm] = derived[nm] + np.abs(col[nm].reshape(P, -1)).max(axis=1)
    for nm, x in got.items():  # (C)
        want = np.asarray(gref[nm])
        if want.size == 0:
            continue
        x = _np(x).reshape(want.shape)
        if np.abs(want).max() == 0:
            assert np.abs(x).max() == 0, (tag, nm)
            continue
        worst[nm] = e = rel_err(x, want)
        assert e < (tol if clean is None or nm == "dL_dtau_sum" and clean.all() else GRAD_TOL_FLIPPED), (tag, nm, e)
        if clean is not None and nm != "dL_dtau_sum" and clean.any():
            worst[nm + "/clean"] = e = float(np.abs(x[clean].astype(np.float64) - want[clean]).max() / (np.abs(want).max() + 1e-30))
            # (no flips, but the sums' conditioning still reaches these rows: what (A) allows them -- MASS_TOL, COND_K -- carried
            # through the chain; next to `tol` only where dL/dalpha cancels, e.g. a map painted in one colour, tools/fuzz_uniform.py)
            cond = float(np.max(derived[nm][clean]) / (np.abs(want).max() + 1e-30)) if derived is not None and nm in chain_allow else 0.0
            assert e < max(tol, CHAIN_K * chain_noise.get(nm, 0.0)) + cond, (tag, nm, "Gaussians untouched by borderline pixels", e, chain_noise.get(nm), cond)
        if derived is not None and nm in chain_allow:  # every Gaussian, flipped or not, against ITS propagated bound
            P = want.shape[0]
            err = np.abs(x.astype(np.float64) - want).reshape(P, -1).max(axis=1)
            allowed = derived[nm] + chain_allow[nm] + chain_err32[nm] + tol * np.abs(want).max()
            worst[nm + "/err_over_propagated_bound"] = float((err / allowed).max())
            i = int(np.argmax(err / allowed))
            assert err[i] <= allowed[i], (tag, nm, "Gaussian %d: end-to-end error %.3e above the compositor bounds carried through the chain %.3e "
                                          "+ chain allowance %.3e + oracle fp32 chain error %.3e" % (i, err[i], derived[nm][i], chain_allow[nm][i], chain_err32[nm][i]))
    if derived is not None and "dL_dtau_sum" in got and "dL_dtau" in chain_allow:
        want = np.asarray(gref["dL_dtau_sum"], np.float64)
        err = np.abs(_np(got["dL_dtau_sum"]).astype(np.float64).reshape(-1) - want.reshape(-1)).max()
        allowed = float((derived["dL_dtau"] + chain_allow["dL_dtau"] + chain_err32["dL_dtau"]).sum()) + tol * np.abs(want).max()
        worst["dL_dtau_sum/err_over_propagated_bound"] = float(err / allowed)
        assert err <= allowed, (tag, "dL_dtau_sum", err, allowed)
    em = gref.get("error_model")
    if em is not None and all(k in got for k in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_ddepth")):  # (A)
        P = gref["dL_dopacity"].shape[0]
        have, want = compositor_sums(got, P), compositor_sums(gref, P)
        mass, cond, flip = (em[k].astype(np.float64) for k in ("term_mass", "cond_slack", "flip_budget"))
        err = np.abs(have - want)
        bound = MASS_TOL * mass + COND_K * cond + FLIP_K * flip + 1e-9 * np.abs(want).max(axis=0, keepdims=True) + 1e-37
        ratio = err / bound
        worst["compositor/err_over_bound"] = float(ratio.max())
        worst["compositor/err_over_mass_p999"] = float(np.quantile(err / (mass + 1e-30), 0.999))
        worst["compositor/needed_flip_budget"] = int((err > MASS_TOL * mass + COND_K * cond + 1e-9 * np.abs(want).max(axis=0, keepdims=True) + 1e-37).any(axis=1).sum())
        i, c = np.unravel_index(np.argmax(ratio), ratio.shape)
        assert ratio.max() < 1.0, (tag, "compositor sum %d of Gaussian %d: got %.6e want %.6e; sum|terms| %.3e cond %.3e flip %.3e"
                                   % (c, i, have[i, c], want[i, c], mass[i, c], cond[i, c], flip[i, c]))
        assert np.abs(have[mass == 0]).max(initial=0.0) == 0.0, (tag, "non-zero sum where no pixel contributes")
    _errlog(tag, **worst)
    return worst


def check_backward(cam, deg, out, args, st, dLc, dLd, tag, **kw):
    """HIP backward vs the oracle's (with term masses) on the same seeds -> (device tuple, oracle dict)."""
    from oracle import oracle as orc

    gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
    gref["error_model"] = orc.error_model(st, dLc, dLd, BORDER_REL, BORDER_REL_T)
    g = gpu_backward(cam, deg, out, args, dLc, dLd)
    assert_grads_close(g, gref, tag, st=st, projmatrix_raw=cam["projmatrix_raw"], **kw)
    return g, gref


def _alpha_walk(st, px, py):
    f = np.float32
    gx = (st["W"] + 15) // 16
    beg, end = st["ranges"][(py // 16) * gx + px // 16]
    ids = st["point_list"][beg:end].astype(np.int64)
    m, co = st["means2D"][ids], st["conic_opacity"][ids]
    dx, dy = m[:, 0] - f(px), m[:, 1] - f(py)
    power = f(-0.5) * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
    mag = 0.5 * (np.abs(co[:, 0]) * dx * dx + np.abs(co[:, 2]) * dy * dy) + np.abs(co[:, 1] * dx * dy)
    alpha = np.minimum(f(0.99), co[:, 3] * np.exp(np.minimum(power, f(0))))
    return ids, power.astype(np.float64), alpha.astype(np.float64), mag.astype(np.float64)


def borderline_pixel(st, px, py, rel=BORDER_REL, rel_T=BORDER_REL_T):
    """True if the oracle's walk of pixel (px, py) meets a contributor that sits on a cut-off of forward.cu:406-535 to
    within the rounding of ANY fp32 evaluation of it: alpha = o exp(power) with power = -(a dx^2 + c dy^2)/2 - b dx dy carries
    an absolute error of a few ulp of the LARGEST of its three products (they cancel for elongated, rotated Gaussians), i.e.
    a relative error of alpha of ~4 eps * mag, mag = (|a| dx^2 + |c| dy^2)/2 + |b dx dy|.  Borderline: alpha within
    rel + 4 eps mag of 1/255; T(1-alpha) within rel_T (+ the accumulated alpha errors of the nearer contributors) of 1e-4;
    power within that absolute error of 0 (power > 0 is skipped).  Only such pixels may legitimately differ between two
    correct fp32 evaluations."""
    ids, power, alpha, mag = _alpha_walk(st, px, py)
    eps = 2.0 ** -23
    thr = 1.0 / 255.0
    T, t_rel = 1.0, rel_T
    for k in range(ids.shape[0]):
        a_rel = rel + 4.0 * eps * mag[k]
        if abs(power[k]) <= 4.0 * eps * mag[k] + 1e-7:
            return True
        if power[k] > 0:
            continue
        a = alpha[k]
        if abs(a - thr) <= a_rel * thr:
            return True
        if a < thr:
            continue
        t = T * (1.0 - a)
        t_rel += a_rel * a / (1.0 - a)
        if abs(t - 1e-4) <= t_rel * 1e-4:
            return True
        if t < 1e-4:
            break
        T = t
    return False


def assert_image_close(got, want, tol, flip_fraction=5e-5, flip_bound=0.02, st=None, tag=None, border_mask=None):
    """Images agree to `tol` (relative to the array's max).  Pixels beyond `tol` are tolerated only if (a) the oracle's own
    walk of that pixel meets a cut-off borderline contributor (`st` = oracle state: checked pixel by pixel), (b) there are at
    most `flip_fraction` of them, and (c) they stay within `flip_bound` (one contributor at the alpha threshold moves a pixel
    by < 1/255).  Without `st` only (b) and (c) can be checked."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got.reshape(want.shape) - want) / scale
    bad = err > tol
    H, W = want.shape[-2], want.shape[-1]
    bad_px = bad.reshape(-1, H, W).any(axis=0)
    n_bad = int(bad_px.sum())
    _errlog(tag or "image", max_err=err.max(), n_bad=n_bad, max_ok=(err[~bad].max() if (~bad).any() else 0.0))
    # (at least one pixel is always allowed: small images would otherwise have a budget of zero)
    assert n_bad <= max(1.0, flip_fraction * H * W), (tag, n_bad, err.max())
    assert err.max() <= flip_bound, (tag, err.max())
    if border_mask is not None:
        assert not (bad_px & ~border_mask).any(), (tag, "%d pixels differ with no borderline contributor" % (bad_px & ~border_mask).sum())
    elif st is not None:
        for py, px in zip(*np.nonzero(bad_px)):
            assert borderline_pixel(st, int(px), int(py)), (tag, "pixel (%d,%d) differs by %.2e with no borderline contributor" % (px, py, err.reshape(-1, H, W)[:, py, px].max()))


def assert_counts_close(got, want, st, tag=None, flip_fraction=2e-4):
    """n_contrib: bit-exact except at pixels with a cut-off borderline contributor (each one verified)."""
    got = np.asarray(got).astype(np.int64).reshape(st["H"], st["W"])
    want = np.asarray(want).astype(np.int64).reshape(st["H"], st["W"])
    ys, xs = np.nonzero(got != want)
    _errlog((tag or "n_contrib"), n_diff=int(ys.size))
    assert ys.size <= max(1.0, flip_fraction * got.size), (tag, ys.size)
    for py, px in zip(ys, xs):
        assert borderline_pixel(st, int(px), int(py)), (tag, "n_contrib of pixel (%d,%d): %d vs %d, no borderline contributor" % (px, py, got[py, px], want[py, px]))


def assert_touched_close(got, want, st, tag=None, rel=BORDER_REL, rel_T=BORDER_REL_T, max_gaussians=2000):
    """n_touched [P] (forward.cu:512-514: pixels where the Gaussian contributes while T(1 - alpha) > 0.5) is integer output, but a
    count of fp32 decisions: it may differ from the oracle's only by pixels whose decision for THAT (pixel, entry) sits on a
    threshold to within the rounding of any fp32 evaluation -- T(1 - alpha) within its accumulated relative uncertainty of 0.5,
    or a cut-off borderline contributor earlier in the pixel's walk (the same criteria as borderline_pixel).  For every Gaussian
    whose count differs, the pixels of its tiles are walked up to its entry and the uncertain ones counted: the difference must
    not exceed them."""
    got = np.asarray(got).astype(np.int64).reshape(-1)
    want = np.asarray(want).astype(np.int64).reshape(-1)
    ids = np.nonzero(got != want)[0]
    _errlog((tag or "n_touched"), n_diff=int(ids.size), sum_abs=int(np.abs(got - want).sum()))
    assert ids.size <= max_gaussians, (tag, "n_touched differs for %d Gaussians" % ids.size)
    if ids.size == 0:
        return 0
    W, H = st["W"], st["H"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    f, eps, thr = np.float32, 2.0 ** -23, 1.0 / 255.0
    pl, ranges, m2, co = st["point_list"], st["ranges"], st["means2D"], st["conic_opacity"]
    for gid in ids:
        r = int(st["radii"][gid])
        mx, my = m2[gid]
        # (the Gaussian's tile rectangle, one tile wider on every side: tiles whose list does not hold it are skipped below)
        x0, x1 = min(gx, max(0, int((mx - r) / 16) - 1)), min(gx, max(0, int((mx + r + 15) / 16) + 1))
        y0, y1 = min(gy, max(0, int((my - r) / 16) - 1)), min(gy, max(0, int((my + r + 15) / 16) + 1))
        uncertain = 0
        for ty in range(y0, y1):
            for tx in range(x0, x1):
                beg, end = ranges[ty * gx + tx]
                lst = pl[beg:end].astype(np.int64)
                pos = np.nonzero(lst == gid)[0]
                if pos.size == 0:
                    continue
                ent = lst[: int(pos[0]) + 1]
                px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0).reshape(-1)
                py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1).reshape(-1)
                ok_px = (px < W) & (py < H)
                dx = m2[ent, 0][None, :] - px[:, None].astype(f)
                dy = m2[ent, 1][None, :] - py[:, None].astype(f)
                a_, b_, c_, o_ = (co[ent, k][None, :] for k in range(4))
                power = (f(-0.5) * (a_ * dx * dx + c_ * dy * dy) - b_ * dx * dy).astype(np.float64)
                mag = (0.5 * (np.abs(a_) * dx * dx + np.abs(c_) * dy * dy) + np.abs(b_ * dx * dy)).astype(np.float64)
                alpha = np.minimum(0.99, o_.astype(np.float64) * np.exp(np.minimum(power, 0.0)))
                a_rel = rel + 4.0 * eps * mag
                contrib = (power <= 0) & (alpha >= thr)
                one_m = np.where(contrib, 1.0 - alpha, 1.0)
                T_after = np.cumprod(one_m, axis=1)          # T(1 - alpha) of entry k where it contributes
                stop = contrib & (T_after < 1e-4)             # the walk ends BEFORE the first such entry
                alive = np.cumsum(stop, axis=1) == 0          # entries the walk still reaches (and that one included: it is tested)
                reached = np.concatenate([np.ones((alive.shape[0], 1), bool), alive[:, :-1]], axis=1)
                t_rel = rel_T + np.cumsum(np.where(contrib, a_rel * alpha / np.maximum(1.0 - alpha, 1e-6), 0.0), axis=1)
                near_alpha = reached & (np.abs(alpha - thr) <= a_rel * thr)
                near_pow = reached & (np.abs(power) <= 4.0 * eps * mag + 1e-7)
                near_stop = reached & contrib & (np.abs(T_after - 1e-4) <= t_rel * 1e-4)
                border = (near_alpha | near_pow | near_stop).any(axis=1)
                k = ent.shape[0] - 1
                near_half = reached[:, k] & (np.abs(T_after[:, k] - 0.5) <= 0.5 * t_rel[:, k]) & (alpha[:, k] >= thr * (1.0 - a_rel[:, k]))
                uncertain += int(((border | near_half) & ok_px).sum())
        assert abs(int(got[gid] - want[gid])) <= uncertain, (tag, "n_touched of Gaussian %d: %d vs %d, but only %d of its pixels have a "
                                                             "decision within rounding of a threshold" % (gid, got[gid], want[gid], uncertain))
    return int(ids.size)


def view_grads_from_sums(sums12, tau_rows, tau_sum):
    """The per-view outputs of the batched backward in GRAD_NAMES order (None where the batched path has no per-view output):
    sums12 [P,12] = BatchContext.view_sums(v) (the reverse compositor's 10 sums per Gaussian: mean2D x y | conic a b c | opacity |
    colour r g b | depth | 2 pads), tau_rows [P,6], tau_sum [6]."""
    s = np.asarray(sums12, np.float32)
    P = s.shape[0]
    m2 = np.concatenate([s[:, 0:2], np.zeros((P, 1), np.float32)], axis=1)
    conic = np.stack([s[:, 2], s[:, 3], np.zeros(P, np.float32), s[:, 4]], axis=1).reshape(P, 2, 2)
    return (m2, s[:, 6:9].copy(), s[:, 5:6].copy(), None, None, None, None, None, np.asarray(tau_rows), np.asarray(tau_sum), conic, s[:, 9:10].copy())
