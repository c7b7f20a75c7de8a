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


@pytest.mark.parametrize("name,naive", [("naive_N4_12x9.npz", False), ("naive_N4_12x9.npz", True), ("naive_edge_N5_12x9.npz", True)])
def test_naive_loop_goldens_on_the_device(golden_dir, name, naive):
    """Loss_Derivative_wrt_mu_and_cov.compute_gradients_2D (the O(HWN^2) loop) reproduced by the dense kernel: the plain golden
    in both modes (no edge case occurs in it), and the edge golden -- alpha >= 0.999 and abs(alpha) < 1e-8 entries -- in the
    naive-guard mode (GSAJ_DENSE_NAIVE_GUARDS)."""
    from gsaj import dense

    g = np.load(os.path.join(golden_dir, name))
    order = np.argsort(g["depth"], kind="stable")
    mu, S, _, _ = dense.compute_gradients_2D(g["mean_2D"][order], g["cov_2D"][order], g["color"][order], g["depth"][order],
                                             g["alpha"][order], g["seed_color"].astype(np.float32), g["seed_depth"].astype(np.float32),
                                             naive_guards=naive)
    inv = np.argsort(order)
    m_mu, m_S = np.abs(g["grad_mu"]).max(), np.abs(g["grad_Sigma"]).max()
    assert np.abs(mu.cpu().numpy()[inv] - g["grad_mu"]).max() < 3e-6 * m_mu
    assert np.abs(S.cpu().numpy()[inv] - g["grad_Sigma"]).max() < 3e-6 * m_S
    if name.startswith("naive_edge"):
        assert np.all(mu.cpu().numpy()[inv][3] == 0)  # the 1e-9-opacity entry is skipped entirely, as in the loop


@pytest.mark.parametrize("name", ["dense_normalised_N15_64x48.npz", "dense_normalised_N64_64x48.npz"])
def test_normalised_coordinate_variant_on_the_device(golden_dir, name):
    """a3, second producer: Loss_Derivative_script.py:820-979 (normalised image coordinates; GSAJ_DENSE_NORMALISED_COORDS) against
    the fixtures that function produced, and against the oracle on the same inputs."""
    from gsaj import dense
    from oracle import dense_oracle as dor

    g = np.load(os.path.join(golden_dir, name))
    intr = (float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]))
    args = (g["mean_2D"], g["cov_2D"], g["color"], g["depth"], g["alpha"], g["seed_color"].astype(np.float32),
            g["seed_depth"].astype(np.float32))
    mu, S, z, c = (x.cpu().numpy() for x in dense.compute_gradients_2D(*args, normalised_intrinsics=intr))
    m_mu, m_S = np.abs(g["grad_mu"]).max(), np.abs(g["grad_Sigma"]).max()
    assert np.abs(mu - g["grad_mu"]).max() < 2e-5 * m_mu
    assert np.abs(S - g["grad_Sigma"]).max() < 2e-5 * m_S
    o_mu, o_S, o_z, o_c = dor.dense_backward(*args, normalised_intrinsics=intr)
    for got, want in ((mu, o_mu), (S, o_S), (z, o_z), (c, o_c)):
        assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    with pytest.raises(Exception, match="NORMALISED_COORDS"):
        dense.compute_gradients_2D(*args, normalised_intrinsics=(0.0, 1.0, 0.0, 0.0))


def test_dense_render_reference_golden(golden_dir):
    """a10: rendered_Image_from_Projected_Gaussians_vectorized (compare.py:973-1018) -- the clipped image the reference hands to
    plt.imshow, recorded by tests/golden/make_goldens_r2.py (every 4th row / column + row and column sums of the full image)."""
    from gsaj import dense

    g = np.load(os.path.join(golden_dir, "dense_N15_640x480.npz"))
    r = np.load(os.path.join(golden_dir, "dense_render_N15_640x480.npz"))
    o = g["order"]
    img, _ = dense.render_projected(g["mean_2D"], g["cov_2D"], g["color"], g["depth"], g["opacities"][o, 0], 480, 640)
    img = np.clip(img.cpu().numpy(), 0.0, 1.0)
    assert np.abs(img[::4, ::4] - r["image_sub4"]).max() < 3e-6 * max(float(r["vmax"]), 1.0)
    assert np.abs(img.astype(np.float64).sum(axis=1) - r["row_sum"]).max() < 1e-5 * np.abs(r["row_sum"]).max()
    assert np.abs(img.astype(np.float64).sum(axis=0) - r["col_sum"]).max() < 1e-5 * np.abs(r["col_sum"]).max()


@pytest.mark.parametrize("name", DENSE)
def test_project_and_sort_golden(golden_dir, name):
    """a4 / a5 / a6 on the device: OrderGaussiansByDepth + GetImagePlaneMeanAndCovs (compute_cov2d, ndc2Pix, SH colours, view-space
    depth) against the arrays the reference's functions produced (tests/golden/make_goldens.py)."""
    from gsaj import dense

    g = np.load(os.path.join(golden_dir, name))
    cam = syn.make_camera(g["w2c"], W=int(g["W"]), H=int(g["H"]), fx=float(g["fx"]), fy=float(g["fy"]), cx=float(g["cx"]), cy=float(g["cy"]))
    pr = {k: v.cpu().numpy() for k, v in dense.project_and_sort(g["means3D"], g["cov3D6"], g["shs"], cam, 3).items()}
    np.testing.assert_array_equal(pr["order"], g["order"])
    assert np.allclose(pr["mean_2D"], g["mean_2D"], rtol=1e-6, atol=2e-5)   # the reference's P is fp32, the pixel values ~1e2
    assert np.allclose(pr["cov_2D"], g["cov_2D"], rtol=2e-6, atol=1e-6)
    assert np.allclose(pr["depth"], g["depth"], rtol=1e-7, atol=1e-7)
    assert np.allclose(pr["color"], g["color"], rtol=1e-9, atol=1e-10)
    assert np.allclose(pr["color_raw"], g["color_raw"][g["order"]], rtol=1e-9, atol=1e-10)


def test_depth_order_is_stable_and_keeps_points_behind_the_camera():
    """A.4: the NumPy path sorts ALL Gaussians by camera z (no z <= 0.2 cull) with Python's stable sort: ties keep index order."""
    from gsaj import dense

    cam = syn.fixture_camera(noisy=True, orthonormal=True, W=64, H=48, fx=57.75, fy=57.75, cx=31.5, cy=23.5)
    sc = syn.make_scene(700, 8, cam, z_range=(-1.0, 3.0), margin=0.3)
    m = sc["means3D"].copy()
    m[100:110] = m[5]      # exact ties in depth
    m[400:420] = m[399]
    pr = dense.project_and_sort(m, syn.covariance6(sc["scales"], sc["rotations"]), sc["shs"], cam, 3)
    order, depth = pr["order"].cpu().numpy(), pr["depth"].cpu().numpy()
    z = (cam["viewmatrix"].astype(np.float64).T @ np.concatenate([m.astype(np.float64), np.ones((700, 1))], 1).T)[2]
    np.testing.assert_array_equal(order, np.argsort(z, kind="stable"))
    assert (depth[1:] >= depth[:-1]).all() and (depth < 0.2).sum() > 50 and sorted(order.tolist()) == list(range(700))


@pytest.mark.parametrize("name", ["dense_N15_64x48.npz", "dense_N15_640x480.npz", "dense_N64_64x48.npz"])
def test_jacobian_test_end_to_end(golden_dir, name):
    """gsaj.dense.jacobian_test: from world Gaussians + camera + ground truth to the four arrays of Jacob_test_result/ with no host
    step in between.  The golden seeds are sign patterns (the generator's `physical_seeds`); ground truth images that reproduce
    exactly those signs under l1_seeds are gt = render - seed (colour) and render - seed with gt > 0 (depth)."""
    from gsaj import dense

    g = np.load(os.path.join(golden_dir, name))
    W, H = int(g["W"]), int(g["H"])
    cam = syn.make_camera(g["w2c"], W=W, H=H, fx=float(g["fx"]), fy=float(g["fy"]), cx=float(g["cx"]), cy=float(g["cy"]))
    o = g["order"]
    img, dep = dense.render_projected(g["mean_2D"], g["cov_2D"], g["color"], g["depth"], g["opacities"][o, 0], H, W)
    img, dep = img.cpu().numpy(), dep.cpu().numpy()
    sc_, sd_ = g["seed_color"].astype(np.float32), g["seed_depth"].astype(np.float32)
    mask = (sc_ != 0).any(axis=-1) | (sd_ != 0)           # the generator's seeds are zero outside its own mask
    gt_c = img - sc_                                       # sign(render - gt) = seed, with a margin of 1
    gt_d = np.where(sd_ > 0, 0.5 * dep, np.where(sd_ < 0, 1.5 * dep + 1e-3, dep))   # same for depth, keeping gt > 0
    assert (gt_d[sd_ != 0] > 0).all()
    r = dense.jacobian_test(g["means3D"], g["cov3D6"], g["opacities"], g["shs"], cam, gt_c, gt_d, mask, 3)
    np.testing.assert_array_equal(r["order"].cpu().numpy(), o)
    for key, want in (("grad_mu_I_pixel", g["grad_mu"]), ("grad_Sigma_I_pixel", g["grad_Sigma"]), ("grad_depth_per_gaussian", g["grad_depth"])):
        assert _rel(r[key].cpu().numpy(), want) < TOL, (key, _rel(r[key].cpu().numpy(), want))
    assert _rel(r["dL_dtau"].cpu().numpy(), g["dL_dtau"]) < 2e-3
    assert r["grad_mu_I_pixel"].shape == (int(g["N"]), 2) and r["dL_dtau"].shape == (6,) and r["dL_dtau"].dtype == __import__("torch").float64
