"""GPU parity of the dense analytic path (the reference's CPU/NumPy semantics) against the golden
vectors produced by the reference's own functions, and against the NumPy oracle at a larger N.
Tolerances: fp32 accumulations 5e-4 of the array's max (the reference sums 307 200 fp32 terms per
Gaussian in chunks of 1000; the kernels sum wave -> workgroup -> fp64), fp64 Jacobians 1e-9."""
import os

import numpy as np
import pytest

from gsaj import synthetic as syn
from oracle import dense_oracle as dor

pytestmark = pytest.mark.gpu
DENSE = ["dense_N1_64x48.npz", "dense_N15_64x48.npz", "dense_N15_64x48_ortho.npz", "dense_N64_64x48.npz",
         "dense_N15_640x480.npz"]
TOL = 5e-4


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("name", DENSE)
def test_dense_golden(golden_dir, name):
    from gsaj import dense

    g = np.load(os.path.join(golden_dir, name))
    o = g["order"]
    N, W, H = int(g["N"]), int(g["W"]), int(g["H"])
    mu, S, z, c = dense.compute_gradients_2D(g["mean_2D"], g["cov_2D"], g["color"], g["depth"], g["opacities"][o, 0],
                                              g["seed_color"].astype(np.float32), g["seed_depth"].astype(np.float32))
    for got, want, nm in ((mu, g["grad_mu"], "mu"), (S, g["grad_Sigma"], "Sigma"), (z, g["grad_depth"], "depth"),
                          (c, g["grad_color"], "color")):
        assert _rel(got.cpu().numpy(), want) < TOL, (name, nm, _rel(got.cpu().numpy(), want))
    cam = syn.make_camera(g["w2c"], W=W, H=H, fx=float(g["fx"]), fy=float(g["fy"]), cx=float(g["cx"]), cy=float(g["cy"]))
    xyz_h = np.concatenate([g["means3D"].astype(np.float64), np.ones((N, 1))], 1)
    dmu, dcov = dense.compute_analytical_jacobians_all_gaussians(xyz_h, g["cov3D6"], cam["w2c"], cam["fx"], cam["fy"], W, H)
    assert np.allclose(dmu.cpu().numpy(), g["dmu_dtau"], rtol=1e-9, atol=1e-11)
    assert np.allclose(dcov.cpu().numpy(), g["dcov_dtau"], rtol=1e-8, atol=1e-9)
    # chain rule on the reference's own per-Gaussian gradients -> the reference's dL/dtau
    tau, parts = dense.assemble_dL_dtau(o, g["grad_mu"], g["grad_Sigma"], g["grad_depth"], g["grad_color"], dmu, dcov,
                                        g["means3D"], cam["w2c"], cam["campos"], g["shs"], 3)
    assert np.allclose(tau.cpu().numpy(), g["dL_dtau"], rtol=1e-8, atol=1e-8 * np.abs(g["dL_dtau"]).max())
    # end to end on the GPU: kernel gradients -> dL/dtau within the fp32 tolerance
    tau2, _ = dense.assemble_dL_dtau(o, mu, S, z, c, dmu, dcov, g["means3D"], cam["w2c"], cam["campos"], g["shs"], 3)
    assert _rel(tau2.cpu().numpy(), g["dL_dtau"]) < 2e-3
    assert mu.shape == (N, 2) and S.shape == (N, 2, 2) and z.shape == (N,)  # layout of Jacob_test_result/*.npy


def test_dense_kat_jacobians(golden_dir):
    from gsaj import dense

    g = np.load(os.path.join(golden_dir, "kat_pose_jacobian.npz"))
    for k in range(g["T_cw"].shape[0]):
        S = g["Sigma_w"][k]
        c6 = np.array([[S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]])
        # fx = fy = 1, W = H = 2 -> unit scaling: raw Eq. 3 / Eq. 4 values
        dmu, dcov = dense.compute_analytical_jacobians_all_gaussians(g["mu_w"][k][None], c6, g["T_cw"][k], 1.0, 1.0, 2, 2)
        assert np.allclose(dmu.cpu().numpy()[0], g["dmu"][k], rtol=1e-10, atol=1e-12), k
        assert np.allclose(dcov.cpu().numpy()[0], g["dcov"][k], rtol=1e-9, atol=1e-11), k


def test_dense_large_n_vs_numpy_oracle_and_render():
    """N = 300 (three LDS chunks) on 160x120: kernels vs the NumPy oracle; forward compositor too."""
    from gsaj import dense

    W, H, N = 160, 120, 300
    cam = syn.fixture_camera(noisy=True, orthonormal=True, W=W, H=H, fx=140.0, fy=140.0, cx=79.5, cy=59.5)
    sc = syn.make_scene(N, 31, cam, z_range=(1.0, 4.0), log_scale_range=(np.log(0.02), np.log(0.15)), margin=-0.05)
    cov6 = syn.covariance6(sc["scales"], sc["rotations"])
    m2, c2, dep = dor.project_gaussians(sc["means3D"], cov6, cam["w2c"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H)
    order = np.argsort(dep, kind="stable")
    dirs = dor.view_dirs(sc["means3D"].astype(np.float64), cam["campos"].astype(np.float64))
    col, _ = dor.colors_from_sh(sc["shs"].astype(np.float64), dirs, 3)
    m2, c2, dep, col, op = m2[order], c2[order], dep[order], col[order], sc["opacities"][order, 0]
    rng = np.random.default_rng(5)
    gc = rng.choice([-1.0, 0.0, 1.0], size=(H, W, 3)).astype(np.float32)
    gd = rng.choice([-1.0, 0.0, 1.0], size=(H, W)).astype(np.float32)
    want = dor.dense_backward(m2, c2, col, dep, op, gc, gd)
    got = dense.compute_gradients_2D(m2, c2, col, dep, op, gc, gd)
    for a, b in zip(got, want):
        assert _rel(a.cpu().numpy(), b) < TOL
    img, d = dense.render_projected(m2, c2, col, dep, op, H, W)
    img_ref, d_ref = dor.dense_render(m2, c2, col, dep, op, H, W)
    assert _rel(img.cpu().numpy(), img_ref) < 1e-4 and _rel(d.cpu().numpy(), d_ref) < 1e-4
    # bit-reproducible
    got2 = dense.compute_gradients_2D(m2, c2, col, dep, op, gc, gd)
    for a, b in zip(got, got2):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
