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


def oracle_forward(cam, sc, deg, bg=(0.0, 0.0, 0.0), precomp=False):
    kw = dict(sh_degree=deg)
    if precomp:
        rng = np.random.default_rng(99)
        kw.update(colors_precomp=rng.uniform(0, 1, size=(sc["means3D"].shape[0], 3)).astype(np.float32),
                  cov3D_precomp=syn.covariance6(sc["scales"], sc["rotations"]))
    else:
        kw.update(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    return orc.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                       cam["tanfovx"], cam["tanfovy"], cam["W"], cam["H"], np.asarray(bg, np.float32), **kw), kw


def seeds(cam, seed=0):
    rng = np.random.default_rng(seed)
    H, W = cam["H"], cam["W"]
    return (rng.normal(size=(3, H, W)).astype(np.float32) / (3 * H * W),
            rng.normal(size=(1, H, W)).astype(np.float32) / (H * W))


def gpu_forward(cam, sc, deg, bg=(0.0, 0.0, 0.0), kw=None, device="cuda:0"):
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
                                False, False)
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


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


def assert_image_close(got, want, tol, flip_fraction=5e-5, flip_bound=0.02):
    """Images agree to `tol` (relative to the array's max) except for at most `flip_fraction` of the
    pixels, where one borderline contributor (alpha within an ulp of 1/255, or T(1-alpha) of 1e-4)
    may fall on the other side of a cut-off under v_exp_f32 vs libm expf; those stay within
    `flip_bound` (one contributor at the alpha threshold changes a pixel by < 1/255)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want) / scale
    # (at least one pixel -- three channel values -- is always allowed: small images would otherwise have a budget of zero)
    assert (err > tol).sum() <= max(3.0, flip_fraction * err.size), ((err > tol).sum(), err.max())
    assert err.max() <= flip_bound, err.max()
