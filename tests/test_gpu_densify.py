"""GPU: densification / pruning bookkeeping (gsaj_densification_stats) against the golden produced by the
reference's own add_densification_stats (gaussian_model.py:767-771) and the mapping loop's max_radii2D / n_obs lines
(slam_backend.py:113-121, 236-250) on CPU tensors (tests/golden/make_densify_goldens.py), and end to end behind a batched
backward: BatchContext -> GaussianModel.densification_step."""
import os

import numpy as np
import pytest

import helpers as hp
from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu


def test_densification_stats_golden(golden_dir):
    import torch
    from gaussian_splatting.scene.gaussian_model import GaussianModel

    g = np.load(os.path.join(golden_dir, "densify_K4_P300.npz"))
    dev = torch.device("cuda:0")
    K, P = g["radii"].shape
    m = GaussianModel(3)
    m._xyz = torch.zeros((P, 3), device=dev)
    m._init_aux()
    m.xyz_gradient_accum.copy_(torch.tensor(g["accum0"]))
    m.denom.copy_(torch.tensor(g["denom0"]))
    m.max_radii2D.copy_(torch.tensor(g["maxr0"]))
    n_obs = m.densification_step(torch.tensor(g["grads"], device=dev), torch.tensor(g["radii"], device=dev), torch.tensor(g["n_touched"], device=dev))
    assert np.allclose(m.xyz_gradient_accum.cpu().numpy(), g["accum"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(m.denom.cpu().numpy(), g["denom"])
    np.testing.assert_array_equal(m.max_radii2D.cpu().numpy(), g["maxr"])
    np.testing.assert_array_equal(n_obs.cpu().numpy(), g["n_obs"])
    # the reference's own call shape: one view, viewspace_points.grad + visibility filter
    m2 = GaussianModel(3)
    m2._xyz = torch.zeros((P, 3), device=dev)
    m2._init_aux()
    m2.xyz_gradient_accum.copy_(torch.tensor(g["accum0"]))
    m2.denom.copy_(torch.tensor(g["denom0"]))
    for k in range(K):
        class VS:
            grad = torch.tensor(g["grads"][k], device=dev)
        m2.add_densification_stats(VS, torch.tensor(g["radii"][k] > 0, device=dev))
    assert np.allclose(m2.xyz_gradient_accum.cpu().numpy(), g["accum"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(m2.denom.cpu().numpy(), g["denom"])
    assert float(m2.max_radii2D.abs().max()) == 0.0   # add_densification_stats alone leaves max_radii2D alone


def test_densification_behind_a_batched_backward():
    import torch
    from gaussian_splatting.scene.gaussian_model import GaussianModel
    from gsaj.rasterizer import BatchContext

    cam0, sc, deg = hp.make("p6000_640x480_sh1")
    K = 4
    cams = syn.keyframe_cameras(K, W=cam0["W"], H=cam0["H"], fx=cam0["fx"], fy=cam0["fy"], cx=cam0["cx"], cy=cam0["cy"])
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    model = GaussianModel.from_activated(sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["shs"], sh_degree=3, device=dev)
    model._init_aux()
    bc = BatchContext(K, P, cam0["W"], cam0["H"], M, dev)
    views, projs = t(np.stack([c["viewmatrix"] for c in cams])), t(np.stack([c["projmatrix"] for c in cams]))
    cps, praw, bg = t(np.stack([c["campos"] for c in cams])), t(cams[0]["projmatrix_raw"]), torch.zeros(3, device=dev)
    geo = dict(sh_degree=deg, shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    bc.forward(bg, t(sc["means3D"]), t(sc["opacities"]), views, projs, cps, cam0["tanfovx"], cam0["tanfovy"], **geo)
    seeds = [hp.seeds(c, seed=70 + k) for k, c in enumerate(cams)]
    g = bc.backward(bg, t(sc["means3D"]), views, projs, praw, cps, cam0["tanfovx"], cam0["tanfovy"], t(np.stack([s[0] for s in seeds])),
                    t(np.stack([s[1] for s in seeds])), **geo)
    n_obs = model.densification_step(g["mean2D"], bc.radii, bc.n_touched)
    vis = bc.radii > 0
    want_acc = (torch.linalg.norm(g["mean2D"][:, :, :2], dim=-1) * vis).double().sum(0)
    assert torch.allclose(model.xyz_gradient_accum[:, 0].double(), want_acc, rtol=1e-5, atol=1e-12)
    assert torch.equal(model.denom[:, 0], vis.sum(0).float()) and torch.equal(model.max_radii2D, bc.radii.max(0).values.float())
    assert torch.equal(n_obs, (bc.n_touched > 0).sum(0).int()) and int(n_obs.max()) == K
