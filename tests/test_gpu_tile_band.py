"""GPU: tile-band sharding of ONE frame (gsaj_set_tile_band, gsaj.tile_band_shard; SURVEY 8e partitioning B).

The reference binds every tile of a frame on one device (cuda_rasterizer/rasterizer_impl.cu:224-352); a banded frame renders
only tile rows [begin, end).  What must hold, whatever the bands (they are emulated here on the one GPU of the box, one
FrameContext per "rank"; the collective that adds the shares is covered by tests/test_cpu_tile_band.py over gloo):

  * inside its band a rank's pixels are BIT-identical to the whole-frame render (same tile lists, same order), outside they
    are background / 0 / 0;
  * radii: the whole-frame radius where the Gaussian has a tile in the band, 0 elsewhere; the union is the whole frame's;
  * n_touched and the instance counts add up exactly (integers);
  * every gradient incl. dL/dtau adds up to the whole-frame value to fp32 rounding (sums over pixels, split by band), and
    the summed shares hold the oracle tolerances of tests/helpers.py;
  * a tracking loop whose dL/dtau is the sum of the band shares follows the whole-frame loop."""
import numpy as np
import pytest

import helpers as hp
from gsaj import synthetic as syn
from gsaj import tile_band_shard as tbs

pytestmark = pytest.mark.gpu


def _tensors(cam, sc, precomp=False):
    import torch

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P = sc["means3D"].shape[0]
    if precomp:
        rng = np.random.default_rng(99)
        kw = dict(colors_precomp=t(rng.uniform(0, 1, size=(P, 3))), cov3D_precomp=t(syn.covariance6(sc["scales"], sc["rotations"])))
        M = 0
    else:
        kw = dict(shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
        M = sc["shs"].shape[1]
    a = dict(bg=t(np.array([0.1, 0.2, 0.3])), means=t(sc["means3D"]), opac=t(sc["opacities"]), view=t(cam["viewmatrix"]),
             proj=t(cam["projmatrix"]), praw=t(cam["projmatrix_raw"]), campos=t(cam["campos"]))
    return dev, t, M, kw, a


def _frame(cam, sc, deg, band, dLc, dLd, precomp=False, record_bits=32, sync=True):
    import torch
    from gsaj.rasterizer import FrameContext

    dev, t, M, kw, a = _tensors(cam, sc, precomp)
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    fc = FrameContext(P, W, H, M, dev, has_scales=not precomp, record_bits=record_bits, per_gaussian_tau=True)
    if band is not None:
        fc.set_tile_band(*band)
    fc.forward(a["bg"], a["means"], a["opac"], a["view"], a["proj"], a["campos"], cam["tanfovx"], cam["tanfovy"], sh_degree=deg, **kw)
    if not sync:  # the asynchronous entry point must honour the band as well
        fc.forward(a["bg"], a["means"], a["opac"], a["view"], a["proj"], a["campos"], cam["tanfovx"], cam["tanfovy"], sh_degree=deg,
                   sync=False, **kw)
        fc.status()
    g = fc.backward(a["bg"], a["means"], a["view"], a["proj"], a["praw"], a["campos"], cam["tanfovx"], cam["tanfovy"], t(dLc), t(dLd),
                    sh_degree=deg, **kw)
    torch.cuda.synchronize()
    return fc, g


GRAD_NAMES = ["mean2D", "conic", "opacity", "color", "depth", "mean3D", "cov3D", "sh", "scale", "rot", "tau", "tau_sum"]


def _check_bands(cam, sc, deg, bands, precomp=False, record_bits=32, sync=True, tag="band"):
    import torch

    dLc, dLd = hp.seeds(cam, seed=77)
    whole, gw = _frame(cam, sc, deg, None, dLc, dLd, precomp, record_bits, sync)
    H = cam["H"]
    bgc = torch.tensor([0.1, 0.2, 0.3], device=whole.color.device).view(3, 1, 1)
    total_R, nt_sum, radii_union = 0, torch.zeros_like(whole.n_touched), torch.zeros_like(whole.radii)
    sums = {}
    for band in bands:
        fc, g = _frame(cam, sc, deg, band, dLc, dLd, precomp, record_bits, sync)
        y0, y1 = band[0] * 16, min(H, band[1] * 16)
        assert torch.equal(fc.color[:, y0:y1], whole.color[:, y0:y1]) and torch.equal(fc.depth[:, y0:y1], whole.depth[:, y0:y1])
        assert torch.equal(fc.opacity[:, y0:y1], whole.opacity[:, y0:y1])
        outside = torch.ones(H, dtype=torch.bool, device=fc.color.device)
        outside[y0:y1] = False
        assert torch.equal(fc.color[:, outside], bgc.expand(3, int(outside.sum()), cam["W"]))
        assert float(fc.depth[:, outside].abs().max() if outside.any() else 0) == 0.0
        assert float(fc.opacity[:, outside].abs().max() if outside.any() else 0) == 0.0
        vis = fc.radii > 0
        assert torch.equal(fc.radii[vis], whole.radii[vis])
        radii_union = torch.maximum(radii_union, fc.radii)
        nt_sum += fc.n_touched
        total_R += fc.true_R
        for n in GRAD_NAMES:
            if g.get(n) is not None and g[n].numel():
                sums[n] = sums.get(n, 0) + g[n].double()
        # a Gaussian invisible in the band gets no gradient from it
        assert float(g["mean3D"][~vis].abs().max() if (~vis).any() else 0) == 0.0
    assert torch.equal(radii_union, whole.radii)
    assert torch.equal(nt_sum, whole.n_touched)
    # a Gaussian spanning several bands is binned once per tile either way: instance counts add up
    assert total_R == whole.true_R
    for n, s in sums.items():
        want = gw[n].double()
        scale = float(want.abs().max())
        e = float((s - want).abs().max()) / max(scale, 1e-30)
        hp._errlog(tag + "/sum_of_bands/" + n, err=e)
        # the band shares are the same per-pixel terms grouped by band: the only difference is where the fp32 per-tile /
        # per-Gaussian partial sums are cut (and the chain applied to each share: it is linear in the compositor sums)
        assert e < 1e-5, (n, e)
    return whole, gw, sums


@pytest.mark.parametrize("scene,world", [("p2000_160x120", 2), ("p6000_640x480_sh1", 3), ("p300_behind_64x48", 3), ("n15_640x480", 4)])
def test_band_shares_add_up_to_the_whole_frame(scene, world):
    cam, sc, deg = hp.make(scene)
    _check_bands(cam, sc, deg, tbs.uniform_bands(cam["H"], world), tag="band/" + scene)


def test_bands_async_forward_precomp_and_fp16_records():
    cam, sc, deg = hp.make("p6000_640x480_sh1")
    _check_bands(cam, sc, deg, [(0, 7), (7, 8), (8, 30)], precomp=True, sync=False, tag="band/async_precomp")
    _check_bands(cam, sc, deg, tbs.uniform_bands(cam["H"], 2), record_bits=16, tag="band/rec16")


def test_bands_chunked_sort(monkeypatch):
    """Tile lists longer than the LDS sort capacity (chunks + merge passes) under a band."""
    from gsaj import rasterizer as C

    monkeypatch.setattr(C, "FORCE_CHUNKED_SORT", True)
    cam, sc, deg = hp.make("p2000_160x120")
    _check_bands(cam, sc, deg, [(0, 3), (3, 8)], tag="band/chunked_sort")


def test_full_size_cfg2_bands_balanced_by_work_hold_the_oracle_tolerances():
    """cfg2 (the benchmark workload), 4 bands cut where the Gaussian-pixel interactions balance: the summed shares against
    the ORACLE's whole-frame gradients, same tolerances as the single-GPU frame (tests/test_gpu_full_size.py)."""
    import os

    import torch
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    orc.set_threads(min(16, os.cpu_count() or 1))
    try:
        cam, sc = syn.config_scene("cfg2")
        deg = 3
        (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=(0.1, 0.2, 0.3))
        dLc, dLd = hp.seeds(cam, seed=77)
        probe, _ = _frame(cam, sc, deg, None, dLc, dLd)
        dbg = C.debug_export(probe.P, probe.R, probe.W, probe.H, probe.geom, probe.binning, probe.img)
        work = tbs.row_work(dbg["n_contrib"])
        assert sum(work) == probe.interactions()
        bands = tbs.balanced_bands(work, 4)
        share = [sum(work[b:e]) / sum(work) for b, e in bands]
        assert max(share) < 0.25 + max(work) / sum(work), share  # within one tile row of the ideal quarter
        whole, gw, sums = _check_bands(cam, sc, deg, bands, tag="band/cfg2")
        gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
        gref["error_model"] = orc.error_model(st, dLc, dLd, hp.BORDER_REL, hp.BORDER_REL_T)
        g = tuple(sums[n[3:]].float() if n[3:] in sums else None for n in hp.GRAD_NAMES)
        hp.assert_grads_close(g, gref, "band/cfg2/oracle", st=st, projmatrix_raw=cam["projmatrix_raw"])
    finally:
        orc.set_threads(1)


def test_band_is_kept_until_changed_and_whole_frame_restores_bitwise():
    import torch

    cam, sc, deg = hp.make("p2000_160x120")
    dLc, dLd = hp.seeds(cam, seed=3)
    whole, gw = _frame(cam, sc, deg, None, dLc, dLd)
    dev, t, M, kw, a = _tensors(cam, sc)
    fc, _ = _frame(cam, sc, deg, (2, 5), dLc, dLd)
    first = fc.color.clone()
    fc.forward(a["bg"], a["means"], a["opac"], a["view"], a["proj"], a["campos"], cam["tanfovx"], cam["tanfovy"], sh_degree=deg, **kw)
    assert torch.equal(fc.color, first)  # still banded
    fc.set_tile_band(0, tbs.tile_rows(cam["H"]))
    fc.forward(a["bg"], a["means"], a["opac"], a["view"], a["proj"], a["campos"], cam["tanfovx"], cam["tanfovy"], sh_degree=deg, **kw)
    g = fc.backward(a["bg"], a["means"], a["view"], a["proj"], a["praw"], a["campos"], cam["tanfovx"], cam["tanfovy"], t(dLc), t(dLd),
                    sh_degree=deg, **kw)
    assert torch.equal(fc.color, whole.color) and torch.equal(fc.radii, whole.radii)
    assert torch.equal(g["tau_sum"], gw["tau_sum"]) and torch.equal(g["mean3D"], gw["mean3D"])


def test_band_on_a_batched_window_equals_banded_single_views():
    """BatchContext.set_tile_band: the band lives in each view's block of the image workspace; the batched launches must give
    each view the bits of a banded single-view frame."""
    import torch
    from gsaj.rasterizer import BatchContext, FrameContext

    cam0, sc, deg = hp.make("p6000_640x480_sh1")
    K = 3
    cams = syn.keyframe_cameras(K, W=cam0["W"], H=cam0["H"], fx=cam0["fx"], fy=cam0["fy"], cx=cam0["cx"], cy=cam0["cy"])
    dev, t, M, kw, a = _tensors(cam0, sc)
    P, W, H = sc["means3D"].shape[0], cam0["W"], cam0["H"]
    views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
    band = (9, 21)
    bc = BatchContext(K, P, W, H, M, dev)
    bc.set_tile_band(*band)
    bc.forward(a["bg"], a["means"], a["opac"], views, projs, cps, cam0["tanfovx"], cam0["tanfovy"], sh_degree=deg, **kw)
    seeds = [hp.seeds(cam0, seed=80 + k) for k in range(K)]
    dLc, dLd = t(np.stack([s[0] for s in seeds])), t(np.stack([s[1] for s in seeds]))
    g = bc.backward(a["bg"], a["means"], views, projs, a["praw"], cps, cam0["tanfovx"], cam0["tanfovy"], dLc, dLd, sh_degree=deg, **kw)
    for k in range(K):
        fc = FrameContext(P, W, H, M, dev)
        fc.set_tile_band(*band)
        fc.forward(a["bg"], a["means"], a["opac"], views[k], projs[k], cps[k], cam0["tanfovx"], cam0["tanfovy"], sh_degree=deg, **kw)
        gs = fc.backward(a["bg"], a["means"], views[k], projs[k], a["praw"], cps[k], cam0["tanfovx"], cam0["tanfovy"], dLc[k], dLd[k],
                         sh_degree=deg, **kw)
        assert torch.equal(bc.color[k], fc.color) and torch.equal(bc.radii[k], fc.radii) and torch.equal(bc.n_touched[k], fc.n_touched)
        assert float(bc.opacity[k][:, : band[0] * 16].abs().max()) == 0.0 and float(bc.opacity[k][:, band[1] * 16:].abs().max()) == 0.0
        assert float((g["tau_all"][k] - gs["tau_sum"]).abs().max()) <= 3e-6 * float(gs["tau_sum"].abs().max())
    bc.set_tile_band(0, tbs.tile_rows(H), views=[1])  # view 1 back to the whole frame, the others stay banded
    bc.forward(a["bg"], a["means"], a["opac"], views, projs, cps, cam0["tanfovx"], cam0["tanfovy"], sh_degree=deg, **kw)
    assert float(bc.opacity[1][:, : band[0] * 16].abs().max()) > 0.0 and float(bc.opacity[0][:, : band[0] * 16].abs().max()) == 0.0


def test_band_survives_the_arena_resize_of_a_batched_window():
    """BatchContext.forward(sync=True) re-runs the batch after growing an arena that was too small.  The re-size must not lose the
    views' tile bands (they live in the image workspaces: zeroing those made every view render the whole frame, and a sum
    all-reduce over band-sharded ranks then counted gradients twice): with a tiny initial capacity, the band shares of a window
    still add up to the whole-frame gradients."""
    import torch
    from gsaj.rasterizer import BatchContext

    cam0, sc, deg = hp.make("p6000_640x480_sh1")
    K = 2
    cams = syn.keyframe_cameras(K, W=cam0["W"], H=cam0["H"], fx=cam0["fx"], fy=cam0["fy"], cx=cam0["cx"], cy=cam0["cy"])
    dev, t, M, kw, a = _tensors(cam0, sc)
    P, W, H = sc["means3D"].shape[0], cam0["W"], cam0["H"]
    views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
    seeds = [hp.seeds(cam0, seed=90 + k) for k in range(K)]
    dLc, dLd = t(np.stack([s[0] for s in seeds])), t(np.stack([s[1] for s in seeds]))

    def window(band, capacity):
        bc = BatchContext(K, P, W, H, M, dev)
        if band is not None:
            bc.set_tile_band(*band)
        if capacity:
            bc._size(capacity)  # far too small: every view aborts, forward(sync=True) must grow the arena and re-run
        st = bc.forward(a["bg"], a["means"], a["opac"], views, projs, cps, cam0["tanfovx"], cam0["tanfovy"], sh_degree=deg, **kw)
        assert not any(ab for _, _, ab in st)
        g = bc.backward(a["bg"], a["means"], views, projs, a["praw"], cps, cam0["tanfovx"], cam0["tanfovy"], dLc, dLd, sh_degree=deg, **kw)
        return bc, {n: g[n].clone() for n in ("mean3D", "opacity", "sh", "scale", "rot", "tau_all")}, st

    whole, gw, stw = window(None, 0)
    rows = tbs.tile_rows(H)
    bands = [(0, 11), (11, rows)]
    shares = [window(b, 300) for b in bands]
    for (bc, g, st), (b0, b1) in zip(shares, bands):
        assert bc.capacity > 300  # the re-size happened
        for k in range(K):  # ... and the band is still in force: nothing outside it
            outside = torch.cat([bc.opacity[k][:, : b0 * 16].reshape(-1), bc.opacity[k][:, b1 * 16:].reshape(-1)])
            assert outside.numel() > 0 and float(outside.abs().max()) == 0.0
    assert [sum(s[2][k][0] for s in shares) for k in range(K)] == [stw[k][0] for k in range(K)]  # instance counts add up
    for n in gw:
        tot = sum(s[1][n].double() for s in shares)
        e = float((tot - gw[n].double()).abs().max() / gw[n].double().abs().max())
        assert e < 5e-6, (n, e)


def test_uninitialised_image_workspace_is_not_mistaken_for_a_band():
    """The synchronous entry points do not ask for a zeroed image workspace (the drop-in binding hands over torch.empty
    memory): whatever bytes it holds, the whole frame is rendered."""
    import torch
    from gsaj.rasterizer import FrameContext

    cam, sc, deg = hp.make("p2000_160x120")
    dLc, dLd = hp.seeds(cam, seed=3)
    whole, _ = _frame(cam, sc, deg, None, dLc, dLd)
    dev, t, M, kw, a = _tensors(cam, sc)
    for fill in (0xFF, 0x01, None):
        fc = FrameContext(sc["means3D"].shape[0], cam["W"], cam["H"], M, dev)
        if fill is None:
            fc.img.random_(0, 256)
        else:
            fc.img.fill_(fill)
        fc.forward(a["bg"], a["means"], a["opac"], a["view"], a["proj"], a["campos"], cam["tanfovx"], cam["tanfovy"], sh_degree=deg, **kw)
        assert torch.equal(fc.color, whole.color) and torch.equal(fc.radii, whole.radii)


def test_empty_band_renders_nothing_and_bad_bands_are_refused():
    import torch
    cam, sc, deg = hp.make("p2000_160x120")
    rows = tbs.tile_rows(cam["H"])
    dLc, dLd = hp.seeds(cam, seed=3)
    fc, g = _frame(cam, sc, deg, (rows, rows), dLc, dLd)
    assert fc.true_R == 0 and int(fc.radii.abs().max()) == 0 and float(g["tau_sum"].abs().max()) == 0.0
    assert float(fc.opacity.abs().max()) == 0.0
    for bad in [(0, 0), (3, 2), (-1, 2), (0, rows + 1)]:
        with pytest.raises(Exception, match="gsaj_set_tile_band"):  # invalid arguments raise plain Exceptions, as the reference's binding does
            fc.set_tile_band(*bad)


def test_tracking_loop_on_band_shares_follows_the_whole_frame_loop():
    """10 tracking iterations (render -> tracking loss -> backward -> device Adam + update_pose, reference
    utils/slam_frontend.py:135-193) with the pose gradient assembled from 3 band shares through pack_pose_terms, against the
    same loop on the whole frame."""
    import torch
    from gsaj.losses import LossSeeds, TRACKING
    from gsaj.pose_step import PoseTracker
    from gsaj.rasterizer import FrameContext

    cam, sc, deg = hp.make("p6000_640x480_sh1")
    dev, t, M, kw, a = _tensors(cam, sc)
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    # ground truth: the render from the fixture pose; the loop starts a centimetre away from it
    gt = FrameContext(P, W, H, M, dev)
    gt.forward(a["bg"], a["means"], a["opac"], a["view"], a["proj"], a["campos"], cam["tanfovx"], cam["tanfovy"], sh_degree=deg, **kw)
    gt_color, gt_depth = gt.color.clone(), gt.depth[0].clone()
    w2c = torch.as_tensor(np.ascontiguousarray(cam["viewmatrix"].T), dtype=torch.float32)
    w2c[:3, 3] += torch.tensor([0.01, -0.008, 0.012])

    def loop(bands):
        pose = PoseTracker(w2c, a["praw"], dev)
        ctxs = []
        for b in bands:
            fc = FrameContext(P, W, H, M, dev)
            if b is not None:
                fc.set_tile_band(*b)
            ctxs.append((fc, LossSeeds(W, H, dev)))
        losses = []
        for _ in range(10):
            packed = torch.zeros(tbs.REDUCED_FLOATS, device=dev)
            for fc, ls in ctxs:
                fc.forward(a["bg"], a["means"], a["opac"], pose.viewmatrix, pose.projmatrix, pose.campos, cam["tanfovx"], cam["tanfovy"],
                           sh_degree=deg, **kw)
                L = ls(TRACKING, 0.9, 0.01, fc.color, fc.depth, fc.opacity, gt_color, gt_depth, None, pose.exposure_a, pose.exposure_b)
                g = fc.backward(a["bg"], a["means"], pose.viewmatrix, pose.projmatrix, a["praw"], pose.campos, cam["tanfovx"],
                                cam["tanfovy"], L["dL_dcolor"], L["dL_ddepth"], sh_degree=deg, pose_only=True, **kw)
                packed += tbs.pack_pose_terms(g["tau_sum"], ls.scalars)  # what all-reduce(sum) does across ranks
            packed = tbs.allreduce_pose_terms(packed)  # single process: identity
            pose.step(packed[tbs.TAU], packed[tbs.EXPOSURE_GRADS])
            losses.append(float(packed[tbs.LOSS_TERMS][0]))
        return pose.w2c.clone(), losses

    w_whole, l_whole = loop([None])
    w_band, l_band = loop(tbs.uniform_bands(H, 3))
    assert l_whole[-1] < l_whole[0]
    np.testing.assert_allclose(l_band, l_whole, rtol=2e-5)
    assert float((w_band - w_whole).abs().max()) < 2e-6
