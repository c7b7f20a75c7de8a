"""CPU: finite-difference pin of the tiled (rasteriser-semantics) oracle's backward.

The CUDA side of the reference cannot run here, so the C oracle's closed-form backward is
checked against central differences of its own forward in a regime where the forward is
smooth: one 16x16 tile, 5 wide Gaussians, every alpha inside (1/255, 0.99), T > 1e-4 -- none
of the rasteriser's cut-offs is active, so analytic and numeric derivatives must agree.

Finding recorded here (and in DESIGN.md): with view-dependent colour (SH degree > 0) the
reference's pose Jacobian uses tau[0:3] -= dL/dmean_sh (backward.cu:141-143); the exact
derivative of the camera-centre motion is +R_cw dL/ddir, so the rho components deliberately
do NOT match finite differences for degree > 0.  Parity mode reproduces the reference."""
import copy

import numpy as np

from gsaj import synthetic as syn
from oracle import dense_oracle as dor
from oracle import oracle as orc

W = H = 16
KW = dict(W=W, H=H, fx=16.0, fy=16.0, cx=W / 2, cy=H / 2)  # cx = W/2: the mean2D->tau path is exact only then
EPS = 2e-3


def se3_exp(tau):
    rho, th = tau[:3], tau[3:]
    Wm = dor.hat(th)
    a = np.linalg.norm(th)
    if a < 1e-5:
        R, V = np.eye(3) + Wm + 0.5 * Wm @ Wm, np.eye(3) + 0.5 * Wm + Wm @ Wm / 6
    else:
        R = np.eye(3) + np.sin(a) / a * Wm + (1 - np.cos(a)) / a**2 * Wm @ Wm
        V = np.eye(3) + (1 - np.cos(a)) / a**2 * Wm + (a - np.sin(a)) / a**3 * Wm @ Wm
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ rho
    return T


def setup():
    cam = syn.fixture_camera(noisy=True, orthonormal=True, **KW)
    sc = syn.make_scene(5, 3, cam, z_range=(2.0, 3.0), log_scale_range=(np.log(2.0), np.log(3.0)), sh_coeffs=16,
                        opacity_range=(0.25, 0.35), margin=-0.3)
    rng = np.random.default_rng(0)
    return cam, sc, rng.normal(size=(3, H, W)).astype(np.float32), rng.normal(size=(1, H, W)).astype(np.float32)


def run(cam, sc, wc, wd, deg, tau=None):
    w2c = cam["w2c"] if tau is None else se3_exp(tau) @ cam["w2c"]
    c = syn.make_camera(w2c, **KW)
    out, st = orc.forward(sc["means3D"], sc["opacities"], c["viewmatrix"], c["projmatrix"], c["campos"], c["tanfovx"],
                          c["tanfovy"], W, H, np.array([0.3, 0.2, 0.1]), shs=sc["shs"], scales=sc["scales"],
                          rotations=sc["rotations"], sh_degree=deg)
    L = (out["color"].astype(np.float64) * wc).sum() + (out["depth"].astype(np.float64) * wd).sum()
    return L, st, c


def test_smooth_regime_is_smooth():
    cam, sc, wc, wd = setup()
    _, st, _ = run(cam, sc, wc, wd, 3)
    assert st["n_contrib"].min() == 5 and st["final_T"].min() > 1e-2
    assert st["num_rendered"] == 5


def test_pose_jacobian_fd_degree0():
    cam, sc, wc, wd = setup()
    _, st, c = run(cam, sc, wc, wd, 0)
    tau = orc.backward(st, wc, wd, c["projmatrix_raw"])["dL_dtau_sum"]
    num = np.zeros(6)
    for k in range(6):
        d = np.zeros(6)
        d[k] = EPS
        num[k] = (run(cam, sc, wc, wd, 0, d)[0] - run(cam, sc, wc, wd, 0, -d)[0]) / (2 * EPS)
    assert np.abs(num - tau).max() < 5e-4 * np.abs(num).max(), (tau, num)


def test_pose_jacobian_fd_degree3_theta_only():
    cam, sc, wc, wd = setup()
    _, st, c = run(cam, sc, wc, wd, 3)
    tau = orc.backward(st, wc, wd, c["projmatrix_raw"])["dL_dtau_sum"]
    num = np.zeros(6)
    for k in range(6):
        d = np.zeros(6)
        d[k] = EPS
        num[k] = (run(cam, sc, wc, wd, 3, d)[0] - run(cam, sc, wc, wd, 3, -d)[0]) / (2 * EPS)
    assert np.abs(num[3:] - tau[3:]).max() < 1e-3 * np.abs(num).max()
    assert np.abs(num[:3] - tau[:3]).max() > 1e-2 * np.abs(num).max()  # the upstream SH->rho term (see module docstring)


def test_parameter_gradients_fd():
    cam, sc, wc, wd = setup()
    _, st, c = run(cam, sc, wc, wd, 3)
    g = orc.backward(st, wc, wd, c["projmatrix_raw"])
    for key, name, tol in [("means3D", "dL_dmean3D", 1e-3), ("scales", "dL_dscale", 5e-3), ("rotations", "dL_drot", 2e-3),
                           ("opacities", "dL_dopacity", 1e-3), ("shs", "dL_dsh", 1e-3)]:
        ana = g[name].reshape(sc[key].shape)
        num = np.zeros_like(ana)
        it = np.nditer(sc[key], flags=["multi_index"])
        for _ in it:
            idx = it.multi_index
            s2 = copy.deepcopy(sc)
            s2[key][idx] += EPS
            a = run(cam, s2, wc, wd, 3)[0]
            s2[key][idx] -= 2 * EPS
            num[idx] = (a - run(cam, s2, wc, wd, 3)[0]) / (2 * EPS)
        assert np.abs(num - ana).max() < tol * np.abs(num).max(), (name, np.abs(num - ana).max() / np.abs(num).max())


def test_tiled_equals_dense_when_cutoffs_inactive(golden_dir):
    """In the smooth regime the rasteriser's compositing backward and the reference's dense NumPy
    backward (compute_gradients_2D_vectorized_chunked semantics) are the same function:
    dL/dmean2D/(W/2,H/2) = dL/dmu_I and -C (dL/dconic) C = dL/dSigma_I."""
    cam, sc, wc, wd = setup()
    c = syn.make_camera(cam["w2c"], **KW)
    out, st = orc.forward(sc["means3D"], sc["opacities"], c["viewmatrix"], c["projmatrix"], c["campos"], c["tanfovx"],
                          c["tanfovy"], W, H, np.zeros(3), shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"],
                          sh_degree=3)
    g = orc.backward(st, wc, wd, c["projmatrix_raw"])
    order = np.argsort(st["depths"], kind="stable")
    co = st["conic_opacity"][order].astype(np.float64)
    con = np.stack([np.stack([co[:, 0], co[:, 1]], -1), np.stack([co[:, 1], co[:, 2]], -1)], -2)
    cov = np.linalg.inv(con)
    mu, S, z, col = dor.dense_backward(st["means2D"][order], cov, st["rgb"][order], st["depths"][order], co[:, 3],
                                       np.transpose(wc, (1, 2, 0)), wd[0])
    m2 = g["dL_dmean2D"][order][:, :2] / np.array([0.5 * W, 0.5 * H])
    assert np.abs(m2 - mu).max() < 1e-3 * np.abs(mu).max()
    gc = g["dL_dconic"][order].astype(np.float64)
    gsym = np.stack([np.stack([gc[:, 0, 0], gc[:, 0, 1]], -1), np.stack([gc[:, 0, 1], gc[:, 1, 1]], -1)], -2)
    S_from_conic = -con @ gsym @ con
    assert np.abs(S_from_conic - S).max() < 2e-3 * np.abs(S).max()
    assert np.abs(g["dL_ddepth"][order, 0] - z).max() < 1e-3 * np.abs(z).max()
    assert np.abs(g["dL_dcolor"][order] - col).max() < 1e-3 * np.abs(col).max()
