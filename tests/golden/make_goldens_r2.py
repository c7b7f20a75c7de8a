#!/usr/bin/env python3
"""More golden fixtures from the reference's own NumPy functions (build container only; same method as make_goldens.py:
the reference files are parsed, only their top-level `def`s are exec'd, fixtures hold inputs and outputs -- data -- only).

    python tests/golden/make_goldens_r2.py

  dense_render_N15_640x480.npz  rendered_Image_from_Projected_Gaussians_vectorized (compare.py:973-1018) on the projected
                                Gaussians of dense_N15_640x480.npz.  The function returns nothing; it hands its image
                                (clipped to [0,1]) to plt.imshow -- the recording `plt` below keeps that argument.  Stored: every
                                4th row / column plus the row and column sums of the full image.
  jacobian_chain_{ortho,similarity}.npz
                                compute_analytical_jacobians_all_gaussians (compare.py:705-760) for 32 Gaussians seen by a
                                camera with cx = W/2, cy = H/2, all inside the un-clamped region of the EWA projection: the
                                reference-derived d(mu_I)/d(tau), d(Sigma_I)/d(tau) the tiled oracle's dL/dtau chain is
                                contracted against in tests/test_oracle_tau_chain.py.
  naive_edge_N5_12x9.npz        Loss_Derivative_wrt_mu_and_cov.compute_gradients_2D (the O(HWN^2) loop) on a scene that
                                exercises its two edge branches: alpha >= 0.999 (suffix term dropped, wrt.py:75-82) and
                                abs(alpha) < 1e-8 (entry skipped, wrt.py:93-94).
  dense_normalised_N{15,64}_64x48.npz
                                the NORMALISED-coordinate variant of the chunked backward, Loss_Derivative_script.py:820-979
                                (pixel grid (u - cx) / fx, (v - cy) / fy from the module globals cx, fx, cy, fy -- injected here;
                                no mask; returns grad_mu, grad_Sigma only), on the projected Gaussians of dense_N{15,64}_64x48.npz
                                converted to normalised coordinates.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (load_defs, quiet, syn)

syn = mg.syn
REF = mg.REF


class RecordingPlot:
    """Stands in for matplotlib.pyplot inside the reference's function: remembers what imshow was given."""

    def __init__(self):
        self.images = []

    def imshow(self, img, *a, **k):
        self.images.append(np.array(img))

    def __getattr__(self, name):
        return lambda *a, **k: None


def dense_render_fixture():
    g = np.load(os.path.join(HERE, "dense_N15_640x480.npz"))
    plt = RecordingPlot()
    ref = mg.load_defs(os.path.join(REF, "Loss_Derivative_script_compare.py"), extra={"plt": plt})
    o = g["order"]
    gl = [dict(mean_2D=g["mean_2D"][i], cov_2D=g["cov_2D"][i], color=g["color"][i], alpha=g["opacities"][o[i], 0])
          for i in range(int(g["N"]))]
    mg.quiet(ref["rendered_Image_from_Projected_Gaussians_vectorized"], gl)
    assert len(plt.images) == 1 and plt.images[0].shape == (480, 640, 3)
    img = plt.images[0].astype(np.float32)
    # kept small: every 4th row and column of the 480x640 image, plus the row / column sums of the whole image (fp64)
    return dict(image_sub4=img[::4, ::4].copy(), row_sum=img.astype(np.float64).sum(axis=1), col_sum=img.astype(np.float64).sum(axis=0),
                vmax=np.float32(img.max()))


def jacobian_chain_fixture(ref, orthonormal, seed):
    W, H, f = 64, 48, 57.75
    cam = syn.fixture_camera(noisy=True, orthonormal=orthonormal, W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0)
    # inside the image with a margin: |x/z| < 1.3 tan(fov/2) everywhere, so the EWA clamp (forward.cu:84-89) is inactive
    sc = syn.make_scene(32, seed, cam, z_range=(0.8, 2.5), log_scale_range=(np.log(0.03), np.log(0.15)), margin=-0.15,
                        sh_coeffs=1)
    cov6 = syn.covariance6(sc["scales"], sc["rotations"])
    xyz_h = np.concatenate([sc["means3D"].astype(np.float64), np.ones((32, 1))], 1)
    dmu, dcov = mg.quiet(ref["compute_analytical_jacobians_all_gaussians"], xyz_h, cov6, cam["w2c"], f, f, W, H)
    return dict(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, w2c=cam["w2c"], means3D=sc["means3D"], scales=sc["scales"],
                rotations=sc["rotations"], opacities=sc["opacities"], shs=sc["shs"], cov3D6=cov6, dmu_dtau=np.asarray(dmu),
                dcov_dtau=np.asarray(dcov))


def naive_edge_fixture(ref_wrt):
    rng = np.random.default_rng(11)
    H, W, N = 9, 12, 5
    gs = []
    for i in range(N):
        A = rng.normal(size=(2, 2))
        g = dict(mean_2D=np.array([rng.uniform(1, W - 2), rng.uniform(1, H - 2)]), cov_2D=A @ A.T + 2.0 * np.eye(2),
                 alpha=rng.uniform(0.3, 0.9), color=rng.uniform(0, 1, 3), depth=1.0 + i)
        gs.append(g)
    # entry 1: opacity 1 and a very wide footprint -> alpha = exp(-|D|^2 / 6000) >= 0.999 within 2.4 px of its mean (where
    # D != 0, so the entry's own dalpha/dmu, dalpha/dSigma do not vanish) and < 0.999 further out
    gs[1]["mean_2D"] = np.array([5.3, 4.2])
    gs[1]["cov_2D"] = 3000.0 * np.eye(2)
    gs[1]["alpha"] = 1.0
    # entry 3: opacity 1e-9 -> abs(alpha) < 1e-8 at every pixel
    gs[3]["alpha"] = 1e-9
    for g in gs:
        g.update(mu_I=g["mean_2D"], Sigma_I=g["cov_2D"], opacity=g["alpha"])
    rc = rng.normal(size=(H, W, 3))
    rd = rng.normal(size=(H, W))
    g_mu, g_S = mg.quiet(ref_wrt["compute_gradients_2D"], gs, rc, rd, np.zeros((H, W, 3)), np.zeros((H, W)), (H, W))
    return dict(H=H, W=W, mean_2D=np.array([g["mean_2D"] for g in gs]), cov_2D=np.array([g["cov_2D"] for g in gs]),
                alpha=np.array([g["alpha"] for g in gs]), color=np.array([g["color"] for g in gs]),
                depth=np.array([g["depth"] for g in gs]), seed_color=np.sign(rc), seed_depth=np.sign(rd),
                grad_mu=np.array(g_mu), grad_Sigma=np.array(g_S))


def dense_normalised_fixture(name, seed):
    g = np.load(os.path.join(HERE, name + ".npz"))
    fx, fy, cx, cy = (float(g[k]) for k in ("fx", "fy", "cx", "cy"))
    H, W, N = int(g["H"]), int(g["W"]), int(g["N"])
    ref = mg.load_defs(os.path.join(REF, "Loss_Derivative_script.py"), extra=dict(fx=fx, fy=fy, cx=cx, cy=cy))
    o = g["order"]
    mean_n = (g["mean_2D"] - np.array([cx, cy])) / np.array([fx, fy])
    cov_n = g["cov_2D"] / np.array([[fx * fx, fx * fy], [fx * fy, fy * fy]])
    gl = [dict(mean_2D=mean_n[i], cov_2D=cov_n[i], color=g["color"][i], depth=g["depth"][i], alpha=g["opacities"][o[i], 0])
          for i in range(N)]
    rng = np.random.default_rng(seed)
    rc, rd = rng.normal(size=(H, W, 3)), rng.normal(size=(H, W))
    g_mu, g_S = mg.quiet(ref["compute_gradients_2D_vectorized_chunked"], gl, rc, rd, np.zeros((H, W, 3)), np.zeros((H, W)), (H, W),
                         chunk_size=500)
    return dict(H=H, W=W, N=N, fx=fx, fy=fy, cx=cx, cy=cy, mean_2D=mean_n, cov_2D=cov_n, color=np.asarray(g["color"]),
                depth=np.asarray(g["depth"]), alpha=np.array([d["alpha"] for d in gl]), seed_color=np.sign(rc).astype(np.int8),
                seed_depth=np.sign(rd).astype(np.int8), grad_mu=np.asarray(g_mu), grad_Sigma=np.asarray(g_S))


def main():
    ref = mg.load_defs(os.path.join(REF, "Loss_Derivative_script_compare.py"))
    ref_wrt = mg.load_defs(os.path.join(REF, "Loss_Derivative_wrt_mu_and_cov.py"))
    np.savez_compressed(os.path.join(HERE, "dense_render_N15_640x480.npz"), **dense_render_fixture())
    np.savez_compressed(os.path.join(HERE, "jacobian_chain_ortho.npz"), **jacobian_chain_fixture(ref, True, 321))
    np.savez_compressed(os.path.join(HERE, "jacobian_chain_similarity.npz"), **jacobian_chain_fixture(ref, False, 322))
    np.savez_compressed(os.path.join(HERE, "naive_edge_N5_12x9.npz"), **naive_edge_fixture(ref_wrt))
    np.savez_compressed(os.path.join(HERE, "dense_normalised_N15_64x48.npz"), **dense_normalised_fixture("dense_N15_64x48", 15))
    np.savez_compressed(os.path.join(HERE, "dense_normalised_N64_64x48.npz"), **dense_normalised_fixture("dense_N64_64x48", 64))
    for n in ("dense_render_N15_640x480", "jacobian_chain_ortho", "jacobian_chain_similarity", "naive_edge_N5_12x9",
              "dense_normalised_N15_64x48", "dense_normalised_N64_64x48"):
        print(n, os.path.getsize(os.path.join(HERE, n + ".npz")), "bytes")


if __name__ == "__main__":
    main()
