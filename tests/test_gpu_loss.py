"""GPU: gsaj_loss_seeds (one-pass losses + pixel-gradient seeds) against the oracle and the reference-generated goldens."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss_seed*_64x48.npz")))
KINDS = {"tracking": 1, "mapping": 0, "mapping_init": 4}


def _run(flags, g, W, H, want_op=False):
    import torch
    from gsaj import losses

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    ls = losses.LossSeeds(W, H, dev)
    out = ls(flags, float(g["alpha"]), float(g["rgb_boundary_threshold"]), t(g["image"]), t(g["depth"]), t(g["opacity"]), t(g["gt"]),
             t(g["gt_depth"]), torch.as_tensor(g["grad_mask"], device=dev), t(np.array([g["exposure_a"]])),
             t(np.array([g["exposure_b"]])), want_opacity_grad=want_op)
    out2 = {k: (v.clone() if v is not None else None) for k, v in out.items()}
    out = ls(flags, float(g["alpha"]), float(g["rgb_boundary_threshold"]), t(g["image"]), t(g["depth"]), t(g["opacity"]), t(g["gt"]),
             t(g["gt_depth"]), torch.as_tensor(g["grad_mask"], device=dev), t(np.array([g["exposure_a"]])),
             t(np.array([g["exposure_b"]])), want_opacity_grad=want_op)
    for k in out:  # second launch reuses the ticket: identical bits
        if out[k] is not None:
            assert torch.equal(out[k], out2[k]), k
    return {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_loss_seeds_match_reference_goldens(path, kind):
    g = np.load(path)
    flags = KINDS[kind] | (2 if bool(g["monocular"]) else 0)
    o = _run(flags, g, 64, 48, want_op=True)
    assert abs(float(o["loss"]) - float(g[kind + "_loss"])) < 2e-7 + 1e-6 * abs(float(g[kind + "_loss"]))
    np.testing.assert_allclose(o["dL_dcolor"], g[kind + "_dL_dimage"], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(o["dL_ddepth"], g[kind + "_dL_ddepth"], rtol=1e-5, atol=1e-10)
    if kind == "tracking":
        np.testing.assert_allclose(o["dL_dopacity"], g[kind + "_dL_dopacity"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(float(o["dL_dexposure_a"]), float(g[kind + "_dL_da"][0]), rtol=2e-4, atol=1e-8)
    np.testing.assert_allclose(float(o["dL_dexposure_b"]), float(g[kind + "_dL_db"][0]), rtol=2e-4, atol=1e-8)


def test_loss_seeds_full_frame_vs_oracle():
    """640x480 (ragged last workgroup is exercised by 641x479 below): oracle on the same seeded inputs."""
    from oracle import loss_oracle as lo

    for (W, H, flags) in ((640, 480, 1), (641, 479, 0), (333, 77, 1 | 2)):
        rng = np.random.default_rng(W + H)
        g = dict(image=rng.uniform(0, 1, (3, H, W)).astype(np.float32), depth=rng.uniform(0.5, 4, (1, H, W)).astype(np.float32),
                 opacity=rng.uniform(0.6, 1.0, (1, H, W)).astype(np.float32), alpha=np.float32(0.9),
                 rgb_boundary_threshold=np.float32(0.01), exposure_a=np.float32(-0.05), exposure_b=np.float32(0.01),
                 grad_mask=rng.uniform(size=(1, H, W)) < 0.6)
        g["gt"] = np.clip(g["image"] + rng.normal(0, 0.1, (3, H, W)), 0, 1).astype(np.float32)
        g["gt_depth"] = (g["depth"][0] + rng.normal(0, 0.05, (H, W))).astype(np.float32)
        g["gt_depth"][rng.uniform(size=(H, W)) < 0.1] = 0
        o = _run(flags, g, W, H)
        r = lo.loss_and_seeds(flags, g["image"], g["depth"], g["opacity"], g["gt"], g["gt_depth"], g["grad_mask"], g["exposure_a"],
                              g["exposure_b"], 0.9, 0.01)
        assert abs(float(o["loss"]) - r["loss"]) < 1e-6 * abs(r["loss"]) + 1e-9
        # a residual within an ulp of zero may take the other sign (fma contraction on the device): <= 1e-5 of the pixels
        for got, want in ((o["dL_dcolor"], r["dL_dimage"]), (o["dL_ddepth"], r["dL_ddepth"])):
            bad = ~np.isclose(got, want, rtol=1e-5, atol=1e-12)
            assert bad.mean() <= 1e-5, bad.sum()
        np.testing.assert_allclose(float(o["dL_dexposure_a"]), r["dL_da"], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(float(o["dL_dexposure_b"]), r["dL_db"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("flags_name", ["mapping", "mapping_mono", "tracking", "mapping_init"])
def test_batched_loss_seeds_equal_the_single_view_calls_bit_for_bit(flags_name):
    """gsaj_loss_seeds_batch (the K keyframes of a mapping window in one launch, utils/slam_backend.py:168-232) gives every view
    exactly what gsaj_loss_seeds gives for its slices -- seeds, loss terms, exposure gradients."""
    import torch
    from gsaj.losses import LossSeeds, LossSeedsBatch, MONOCULAR, NO_EXPOSURE, TRACKING

    flags = {"mapping": 0, "mapping_mono": MONOCULAR, "tracking": TRACKING, "mapping_init": NO_EXPOSURE}[flags_name]
    dev = torch.device("cuda:0")
    K, W, H = 5, 100, 75
    g = torch.Generator(device="cpu").manual_seed(7)
    r = lambda *shape: torch.rand(*shape, generator=g).to(dev)  # noqa: E731
    image, depth, opacity = r(K, 3, H, W), r(K, 1, H, W) * 3, r(K, 1, H, W)
    gt_image, gt_depth = r(K, 3, H, W), (r(K, H, W) * 3) * (r(K, H, W) > 0.2)
    mask = r(K, H, W) > 0.3
    ea, eb = (r(K) - 0.5) * 0.2, (r(K) - 0.5) * 0.1
    noexp = bool(flags & NO_EXPOSURE)
    lb = LossSeedsBatch(K, W, H, dev)
    for _ in range(2):  # twice: the tickets must have been reset
        ob = lb(flags, 0.9, 0.01, image, depth, opacity, gt_image, None if flags & MONOCULAR else gt_depth, mask if flags & TRACKING else None,
                None if noexp else ea, None if noexp else eb)
    ls = LossSeeds(W, H, dev)
    for k in range(K):
        o = ls(flags, 0.9, 0.01, image[k], depth[k], opacity[k], gt_image[k], None if flags & MONOCULAR else gt_depth[k].contiguous(),
               mask[k] if flags & TRACKING else None, None if noexp else ea[k:k + 1], None if noexp else eb[k:k + 1])
        assert torch.equal(ob["dL_dcolor"][k], o["dL_dcolor"]) and torch.equal(ob["dL_ddepth"][k], o["dL_ddepth"])
        assert torch.equal(lb.scalars[k], ls.scalars)
    with pytest.raises(Exception, match="gsaj_loss_seeds_batch"):
        lb(8, 0.9, 0.01, image, depth, opacity, gt_image, gt_depth)  # COMPUTE_LOSS has no batched form
