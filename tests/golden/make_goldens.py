#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN NumPy
functions (build container only: /root/reference does not exist on the GPU box).

The reference scripts cannot be imported as modules (their imports need the CUDA
extension, cv2, open3d, and their __main__ needs missing blobs), but every hot-path
function in them is pure NumPy.  This script parses the files and exec's only the
top-level `def` nodes into a namespace holding numpy/math -- nothing of the reference's
source is written anywhere; the fixtures hold inputs and outputs (data) only.

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz, ref_timing.json

Fixtures
  kat_pose_jacobian.npz   GetAnalyticalJcobian on KAT-1/KAT-2 + 24 random cases
  dense_N{1,15,64}_64x48.npz, dense_N15_640x480.npz
                          compute_cov2d / ndc2Pix / compute_colors_from_sh / OrderGaussiansByDepth /
                          compute_gradients_2D_vectorized_chunked / compute_analytical_jacobians_all_gaussians
                          (+ compute_sh_backward_single, dnormvdv per Gaussian) on seeded synthetic scenes
  naive_N4_12x9.npz       Loss_Derivative_wrt_mu_and_cov.compute_gradients_2D (the O(HWN^2) loop)
  ref_timing.json         wall time of the reference's chunked backward (N=15, optionally N=256)
"""
import ast
import contextlib
import io
import json
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
REF = "/root/reference"

from gsaj import synthetic as syn  # noqa: E402
from oracle import dense_oracle as dor  # noqa: E402


class _NoPlot:
    def __getattr__(self, name):
        return lambda *a, **k: None


def load_defs(path, extra=None):
    """exec only the top-level FunctionDef nodes of a reference script."""
    with open(path) as fh:
        tree = ast.parse(fh.read())
    tree.body = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    ns = {"np": np, "math": math, "plt": _NoPlot(), "Dict": dict, "Any": object}
    try:
        import torch
        ns["torch"] = torch
    except Exception:  # annotation only
        pass
    ns.update(extra or {})
    for node in tree.body:  # drop annotations/defaults that need missing names
        node.returns = None
        for a in node.args.args:
            a.annotation = None
    exec(compile(tree, path, "exec"), ns)
    return ns


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def cov6_to_mat(c):
    return np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]], np.float64)


def kat_fixture(ref):
    cases = []
    T1 = np.array([[.8047, -.3106, .5059, 1], [.5059, .8047, -.3106, 1], [-.3106, .5059, .8047, 1], [0, 0, 0, 1.]])
    T2 = np.eye(4)
    T2[:3, 3] = 1.0
    mu = np.array([2., 3., 4., 1.])
    S = np.array([[1., 2., 3.], [2., 4., 5.], [3., 5., 9.]])
    cases += [(T1, mu, S), (T2, mu, S)]
    rng = np.random.default_rng(7)
    for k in range(24):
        q = syn.random_unit_quaternions(rng, 1)[0]
        r, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        if k % 3 == 2:
            R = R * rng.uniform(0.3, 1.5)  # similarity, like the fixture camera
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rng.normal(size=3)
        m = np.append(rng.normal(size=3), 1.0)
        if (T @ m)[2] < 0.3:
            T[2, 3] += 0.5 - (T @ m)[2] + 1.0
        A = rng.normal(size=(3, 3))
        cases.append((T, m, A @ A.T * 0.1))
    out = dict(T_cw=[], mu_w=[], Sigma_w=[], dmu=[], dcov=[])
    for T, m, Sg in cases:
        a, b = quiet(ref["GetAnalyticalJcobian"], T, m, Sg, 1.0, 1.0)
        out["T_cw"].append(T), out["mu_w"].append(m), out["Sigma_w"].append(Sg)
        out["dmu"].append(a), out["dcov"].append(b)
    return {k: np.array(v) for k, v in out.items()}


def physical_seeds(m2, c2, col, dep, opac, cam_gt_args, H, W):
    """Per-pixel L1 sign seeds from a dense render at the noisy pose against one at a
    slightly different (pixel-shifted) configuration: realistic structure, values in {-1,0,1}."""
    img, d = dor.dense_render(m2, c2, col, dep, opac, H, W)
    m2b = m2 + np.array([1.7, -1.1])
    img2, d2 = dor.dense_render(m2b, c2, col * 0.9, dep * 1.02, opac, H, W)
    mask = (img2.sum(-1) > 1e-3) | (img.sum(-1) > 1e-3)
    sc = (np.sign(img - img2) * mask[..., None]).astype(np.int8)
    sd = (np.sign(d - d2) * mask).astype(np.int8)
    return sc, sd


def dense_fixture(ref, N, W, H, seed, cam):
    fx, fy = cam["fx"], cam["fy"]
    z_lo = 0.8 if N <= 15 else 0.6
    ls = (math.log(0.01), math.log(0.05)) if W >= 320 else (math.log(0.02), math.log(0.12))
    sc = syn.make_scene(N, seed, cam, z_range=(z_lo, 1.6), log_scale_range=ls, margin=-0.1)
    cov6 = syn.covariance6(sc["scales"], sc["rotations"])
    xyz = sc["means3D"]
    w2c = cam["w2c"]
    xyz_h = np.concatenate([xyz.astype(np.float64), np.ones((N, 1))], 1)
    xyz_cam = (w2c @ xyz_h.T).T[:, :3]
    order = [i for i, _ in ref["OrderGaussiansByDepth"](xyz_cam)]
    dirs = ref["compute_viewing_directions"](xyz.astype(np.float64), cam["campos"].astype(np.float64))
    colors = ref["compute_colors_from_sh"](sc["shs"].astype(np.float64), dirs, deg=3)
    raw = ref["eval_sh"](3, sc["shs"].astype(np.float64), dirs) + 0.5
    viewmatrix = cam["viewmatrix"]  # W2C^T, float32, as the reference passes it
    projT = cam["projmatrix"].T.astype(np.float64)
    gl = []
    for idx in order:
        mu = xyz[idx]
        ph = projT @ np.array([mu[0], mu[1], mu[2], 1.0])
        pp = ph[:3] / (ph[3] + 0.0000001)
        cov2, _ = quiet(ref["compute_cov2d"], mu, fx, fy, cam["tanfovx"], cam["tanfovy"], cov6[idx], viewmatrix)
        pv = viewmatrix.T @ np.array([mu[0], mu[1], mu[2], 1.0])
        gl.append(dict(idx=idx, mean_2D=np.array([ref["ndc2Pix"](pp[0], W), ref["ndc2Pix"](pp[1], H)]),
                       cov_2D=cov2, color=colors[idx], depth=pv[2], alpha=sc["opacities"][idx][0]))
    m2 = np.array([g["mean_2D"] for g in gl])
    c2 = np.array([g["cov_2D"] for g in gl])
    col = np.array([g["color"] for g in gl])
    dep = np.array([g["depth"] for g in gl])
    opac = np.array([g["alpha"] for g in gl])
    s_c, s_d = physical_seeds(m2, c2, col, dep, opac, None, H, W)
    rendered_color = s_c.astype(np.float32)
    gt_color = np.zeros((H, W, 3), np.float32)
    gt_depth = np.ones((H, W), np.float32)
    rendered_depth = gt_depth + s_d.astype(np.float32)
    mask = np.ones((H, W), bool)
    t0 = time.time()
    g_mu, g_S, g_z, g_c = quiet(ref["compute_gradients_2D_vectorized_chunked"], gl, rendered_color, rendered_depth,
                                gt_color, gt_depth, mask, (H, W), chunk_size=1000)
    t_chunked = time.time() - t0
    dmu_all, dcov_all = quiet(ref["compute_analytical_jacobians_all_gaussians"], xyz_h, cov6, w2c, fx, fy, W, H)
    # dL/dtau: the reference assembles it in module-level code (compare.py:1587-1695) that
    # cannot be imported; restate that loop here on top of the reference's own constituents.
    tau = np.zeros(6)
    sh_terms = np.zeros((N, 3))
    campos = cam["campos"].astype(np.float64)
    clamped = raw < 0.0
    for i, idx in enumerate(order):
        pc = w2c @ xyz_h[idx]
        g = g_c[i].astype(np.float64).copy()
        g[clamped[idx]] = 0.0
        dorig = xyz[idx].astype(np.float64) - campos
        dn = dorig / (np.linalg.norm(dorig) + 1e-8)
        ddir = ref["compute_sh_backward_single"](dn, sh64(sc["shs"][idx]), g, deg=3)
        dmean = ref["dnormvdv"](dorig, ddir)
        sh_terms[i] = dmean
        tau = (tau + g_mu[i] @ dmu_all[idx] + g_S[i].reshape(4) @ dcov_all[idx]
               + g_z[i] * np.array([0, 0, 1, pc[1], -pc[0], 0], np.float64) + np.concatenate([-dmean, np.zeros(3)]))
    fixture = dict(
        N=N, W=W, H=H, seed=seed, fx=fx, fy=fy, cx=cam["cx"], cy=cam["cy"], w2c=w2c,
        means3D=xyz, scales=sc["scales"], rotations=sc["rotations"], opacities=sc["opacities"], shs=sc["shs"], cov3D6=cov6,
        order=np.array(order), mean_2D=m2, cov_2D=c2, color=col, color_raw=raw, depth=dep,
        seed_color=s_c, seed_depth=s_d,
        grad_mu=g_mu, grad_Sigma=g_S, grad_depth=g_z, grad_color=g_c,
        dmu_dtau=dmu_all, dcov_dtau=dcov_all, sh_dmean=sh_terms, dL_dtau=tau)
    # (rendered_Image_from_Projected_Gaussians_vectorized, compare.py:973-1018, only plots: it
    # returns nothing, so the dense forward has no reference output to pin.)
    return fixture, t_chunked


def sh64(a):
    return a.astype(np.float64)


def naive_fixture(ref_wrt):
    rng = np.random.default_rng(3)
    H, W, N = 9, 12, 4
    gs = []
    for i in range(N):
        A = rng.normal(size=(2, 2))
        g = dict(mean_2D=np.array([rng.uniform(1, W - 2), rng.uniform(1, H - 2)]),
                 cov_2D=A @ A.T + 2.0 * np.eye(2), alpha=rng.uniform(0.3, 0.9),
                 color=rng.uniform(0, 1, 3), depth=1.0 + i)
        g.update(mu_I=g["mean_2D"], Sigma_I=g["cov_2D"], opacity=g["alpha"])  # key names of wrt.py:121-144
        gs.append(g)
    rc = rng.normal(size=(H, W, 3))
    rd = rng.normal(size=(H, W))
    g_mu, g_S = quiet(ref_wrt["compute_gradients_2D"], gs, rc, rd, np.zeros((H, W, 3)), np.zeros((H, W)), (H, W))
    return dict(H=H, W=W, mean_2D=np.array([g["mean_2D"] for g in gs]), cov_2D=np.array([g["cov_2D"] for g in gs]),
                alpha=np.array([g["alpha"] for g in gs]), color=np.array([g["color"] for g in gs]),
                depth=np.array([g["depth"] for g in gs]), seed_color=np.sign(rc), seed_depth=np.sign(rd),
                grad_mu=np.array(g_mu), grad_Sigma=np.array(g_S))


def main():
    ref = load_defs(os.path.join(REF, "Loss_Derivative_script_compare.py"))
    ref_wrt = load_defs(os.path.join(REF, "Loss_Derivative_wrt_mu_and_cov.py"))
    np.savez_compressed(os.path.join(HERE, "kat_pose_jacobian.npz"), **kat_fixture(ref))
    np.savez_compressed(os.path.join(HERE, "naive_N4_12x9.npz"), **naive_fixture(ref_wrt))
    timing = {"cores": os.cpu_count(), "numpy": np.__version__, "function": "compute_gradients_2D_vectorized_chunked"}
    small = dict(W=64, H=48, fx=57.75, fy=57.75, cx=31.5, cy=23.5)
    for N, seed in [(1, 1), (15, 15), (64, 64)]:
        for ortho in ([False] if N != 15 else [False, True]):
            cam = syn.fixture_camera(noisy=True, orthonormal=ortho, **small)
            fx_, _ = dense_fixture(ref, N, 64, 48, seed, cam)
            name = "dense_N%d_64x48%s.npz" % (N, "_ortho" if ortho else "")
            np.savez_compressed(os.path.join(HERE, name), **fx_)
    cam = syn.fixture_camera(noisy=True)
    fx_, t = dense_fixture(ref, 15, 640, 480, 15, cam)
    np.savez_compressed(os.path.join(HERE, "dense_N15_640x480.npz"), **fx_)
    timing["N15_640x480_s"] = t
    timing["N15_pairs_per_s"] = 15 * 640 * 480 / t
    if "--time256" in sys.argv:
        _, t = dense_fixture(ref, 256, 640, 480, 256, cam)
        timing["N256_640x480_s"] = t
        timing["N256_pairs_per_s"] = 256 * 640 * 480 / t
    with open(os.path.join(HERE, "ref_timing.json"), "w") as fh:
        json.dump(timing, fh, indent=1)
    print(json.dumps(timing))


if __name__ == "__main__":
    main()
