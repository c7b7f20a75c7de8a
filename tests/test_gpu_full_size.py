"""GPU: HIP path vs the CPU oracle at the FULL size of every BASELINE.json config -- forward structure (integers
bit-exact), images, and EVERY gradient incl. dL/dtau -- plus the size-independent properties (sorted lists,
opacity + final_T = 1, idempotence, linearity of the backward in the pixel seeds).

  cfg2  50 000 Gaussians SH-3, 640x480 (the benchmark workload)          fp32 records
  cfg3  300 000 Gaussians SH-0, 1200x680 (Replica calibration)           fp32 records
  cfg4  100 000 Gaussians SH-3, 640x480 TUM fr1 calibration, 8 keyframes of one map: per-keyframe dL/dtau and the
        SUM over keyframes of the per-Gaussian gradients (what one loss.backward() of the mapping window leaves in .grad,
        reference utils/slam_backend.py:168-232)
  cfg5  1 000 000 Gaussians SH-0, 1280x720                                fp32 records AND fp16-storage records

The oracle runs tile-parallel on the box's host cores (its results do not depend on the thread count).  The fp16-record
runs are compared with the oracle in ITS fp16-record mode (conic / opacity / colour rounded to half once, fp32
everywhere else), so they are held to the same tolerances as fp32; the reference has no fp16 path, the rounding points
are this repository's definition of BASELINE config 5 ("fp16 splat with fp32 Jacobian accumulation").
Real Replica / TUM / EuRoC frames are not available offline: synthetic stand-ins of the same shapes."""
import os

import numpy as np
import pytest

import helpers as hp
from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def oracle_threads():
    from oracle import oracle as orc

    orc.set_threads(min(16, os.cpu_count() or 1))
    yield
    orc.set_threads(1)


def _structure_and_images(C, cam, sc, deg, out, ref, st, tag):
    import torch

    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    assert R == ref["num_rendered"] == int(st["tiles_touched"].sum())
    dbg = C.debug_export(P, R, W, H, geom, binning, img)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    np.testing.assert_array_equal(dbg["tiles_touched"].cpu().numpy(), st["tiles_touched"])
    np.testing.assert_array_equal(dbg["point_list"].cpu().numpy().astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"].cpu().numpy(), st["ranges"])
    # every tile list is sorted by (depth, id)
    pl, rg = dbg["point_list"].long(), dbg["ranges"].long()
    d = dbg["depths"][pl]
    tile_of = torch.repeat_interleave(torch.arange(rg.shape[0], device=pl.device), rg[:, 1] - rg[:, 0])
    same = tile_of[1:] == tile_of[:-1]
    assert bool(((d[1:] > d[:-1]) | ((d[1:] == d[:-1]) & (pl[1:] > pl[:-1])) | ~same).all())
    assert torch.allclose(opacity[0] + dbg["final_T"], torch.ones_like(dbg["final_T"]), atol=1e-6)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    per_pixel_len = (rg[:, 1] - rg[:, 0]).view(gy, gx).repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    assert int((dbg["n_contrib"].long() > per_pixel_len).sum()) == 0
    hp.assert_counts_close(dbg["n_contrib"].cpu().numpy(), st["n_contrib"], st, tag=tag)
    # integer output, but a count of fp32 threshold decisions: every differing Gaussian must have that many pixels whose
    # T(1 - alpha) > 0.5 (or an earlier cut-off) decision lies within rounding of its threshold
    hp.assert_touched_close(n_touched.cpu().numpy(), ref["n_touched"], st, tag=tag + "/n_touched")
    for nm, got, want in (("color", color, ref["color"]), ("depth", depth, ref["depth"]), ("opacity", opacity, ref["opacity"])):
        hp.assert_image_close(got.cpu().numpy(), want, hp.IMG_TOL, st=st, tag=tag + "/" + nm)


FULL = [("cfg2", 32), ("cfg3", 32), ("cfg5", 32), ("cfg5", 16)]


@pytest.mark.parametrize("wl,bits", FULL, ids=["%s_rec%d" % c for c in FULL])
def test_full_size_parity_every_gradient(wl, bits):
    from gsaj import rasterizer as C

    cam, sc = syn.config_scene(wl)
    M = sc["shs"].shape[1]
    deg = int(round(M ** 0.5)) - 1
    tag = "full/%s/rec%d" % (wl, bits)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, record_bits=bits)
    out, args = hp.gpu_forward(cam, sc, deg, kw=kw, record_bits=bits)
    _structure_and_images(C, cam, sc, deg, out, ref, st, tag)
    dLc, dLd = hp.seeds(cam, seed=31)
    g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, tag)
    # idempotence (same inputs, same bits) and linearity of the backward in the seeds
    g2 = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
    for a, b in zip(g, g2):
        assert a is None or bool((a == b).all())
    s2c, s2d = hp.seeds(cam, seed=32)
    gb = hp.gpu_backward(cam, deg, out, args, s2c, s2d)
    gc = hp.gpu_backward(cam, deg, out, args, 2.0 * dLc - 0.5 * s2c, 2.0 * dLd - 0.5 * s2d)
    for nm, a, b, c in zip(hp.GRAD_NAMES, g, gb, gc):
        if a is None or a.numel() == 0:
            continue
        lin = 2.0 * a - 0.5 * b
        assert float((c - lin).abs().max()) <= 2e-5 * float(c.abs().max()) + 1e-30, nm


def test_cfg4_mapping_window_8_keyframes():
    """BASELINE config 4 on ONE GPU: the 8 keyframes of a mapping window over one shared map, one after the other through
    the single-view C ABI; every keyframe's dL/dtau row and the sum over keyframes of the per-Gaussian gradients against the
    oracle's (the multi-GPU run all-reduces exactly this sum, gsaj.keyframe_shard)."""
    from gsaj import rasterizer as C

    cams, sc = syn.config_window("cfg4", 8)
    deg = 3
    sums, sums_ref = {}, {}
    per_gaussian = ["dL_dmean3D", "dL_dopacity", "dL_dsh", "dL_dscale", "dL_drot", "dL_dcov3D"]
    for k, cam in enumerate(cams):
        tag = "cfg4/kf%d" % k
        (ref, st), kw = hp.oracle_forward(cam, sc, deg)
        out, args = hp.gpu_forward(cam, sc, deg, kw=kw)
        if k in (0, 7):
            _structure_and_images(C, cam, sc, deg, out, ref, st, tag)
        else:
            assert out[0] == ref["num_rendered"]
            hp.assert_image_close(out[1].cpu().numpy(), ref["color"], hp.IMG_TOL, st=st, tag=tag + "/color")
        dLc, dLd = hp.seeds(cam, seed=40 + k)
        g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, tag)
        got = dict(zip(hp.GRAD_NAMES, g))
        for nm in per_gaussian:
            sums[nm] = sums.get(nm, 0) + got[nm].double().cpu().numpy().reshape(gref[nm].shape)
            sums_ref[nm] = sums_ref.get(nm, 0) + gref[nm].astype(np.float64)
    worst = {}
    for nm in per_gaussian:
        worst[nm] = hp.rel_err(sums[nm], sums_ref[nm])
        assert worst[nm] < 1e-3, (nm, worst[nm])  # every keyframe's rows are bounded one by one above; the sum inherits their flips
    hp._errlog("cfg4/sum_over_keyframes", **worst)


def test_bench_window_through_the_batched_path_three_layers():
    """The path bench.py times -- cfg2, ONE window of 8 keyframes with syn.keyframe_cameras, the bench's pixel-gradient seeds,
    through BatchContext (gsaj_rasterize_forward_batch / _backward_batch) -- held to the same three layers as the single-view
    path (helpers.assert_grads_close), per view: (A) the reverse compositor's 10 sums per Gaussian (k_render_bwd +
    k_gather_sums; exported per view) inside the oracle's error model, (B) the per-Gaussian chain on the device's own sums
    against the chain carried in fp64 (per-view: the dL/dtau rows and their sum, k_chain_window), (C) end to end; integers
    bit-exact, images and n_contrib / n_touched different only where a threshold decision lies within rounding.  Then the
    window's SUMS over the 8 keyframes (dL/dmean3D, dL/dcov3D via scale / rotation, dL/dSH, dL/dopacity: k_chain_window)
    against the sum of the per-view fp64 chains on the device's sums, every row bounded by the SUM of the per-view bounds of
    layer (B); dL/dopacity by the sum of the per-view bounds of layer (A)."""
    import torch
    from gsaj import rasterizer as C
    from gsaj.rasterizer import BatchContext
    from oracle import oracle as orc

    K = 8
    cam0, sc = syn.config_scene("cfg2")
    cams = syn.keyframe_cameras(K, **{k: cam0[k] for k in ("W", "H", "fx", "fy", "cx", "cy")})
    P, W, H, M = sc["means3D"].shape[0], cam0["W"], cam0["H"], sc["shs"].shape[1]
    deg = 3
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
    rng = np.random.default_rng(1234)  # bench.py, rank 0
    dLc = (rng.normal(size=(K, 3, H, W)) / (3 * H * W)).astype(np.float32)
    dLd = (rng.normal(size=(K, 1, H, W)) / (H * W)).astype(np.float32)
    geo = dict(sh_degree=deg, shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
    bg, praw = torch.zeros(3, device=dev), t(cam0["projmatrix_raw"])
    bc = BatchContext(K, P, W, H, M, dev, per_gaussian_tau=True)
    bc.forward(bg, t(sc["means3D"]), t(sc["opacities"]), views, projs, cps, cam0["tanfovx"], cam0["tanfovy"], sync=True, **geo)
    # the timed region runs asynchronous windows: so does this one (same kernels, no host round trip)
    bc.forward(bg, t(sc["means3D"]), t(sc["opacities"]), views, projs, cps, cam0["tanfovx"], cam0["tanfovy"], sync=False, **geo)
    g = bc.backward(bg, t(sc["means3D"]), views, projs, praw, cps, cam0["tanfovx"], cam0["tanfovy"], t(dLc), t(dLd), **geo)
    stt = bc.status()
    assert not any(ab for _, _, ab in stt)
    f64 = lambda x: x.double().cpu().numpy()  # noqa: E731
    chain_names = ["dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"]
    truth = {n: 0.0 for n in chain_names}
    allowed = {n: 0.0 for n in chain_names}
    op_want, op_bound = 0.0, 0.0
    for k, cam in enumerate(cams):
        tag = "bench_window/kf%d" % k
        (ref, st), kw = hp.oracle_forward(cam, sc, deg)
        assert stt[k][0] == ref["num_rendered"]
        blk = lambda buf, stride: buf[k * stride:(k + 1) * stride]  # noqa: E731
        dbg = C.debug_export(P, bc.capacity, W, H, blk(bc.geom, bc.geom_stride), blk(bc.binning, bc.bin_stride), blk(bc.img, bc.img_stride))
        np.testing.assert_array_equal(bc.radii[k].cpu().numpy(), ref["radii"])
        np.testing.assert_array_equal(dbg["point_list"].cpu().numpy().astype(np.uint32)[:ref["num_rendered"]], st["point_list"])
        np.testing.assert_array_equal(dbg["ranges"].cpu().numpy(), st["ranges"])
        hp.assert_counts_close(dbg["n_contrib"].cpu().numpy(), st["n_contrib"], st, tag=tag)
        hp.assert_touched_close(bc.n_touched[k].cpu().numpy(), ref["n_touched"], st, tag=tag + "/n_touched")
        for nm, got in (("color", bc.color[k]), ("depth", bc.depth[k]), ("opacity", bc.opacity[k])):
            hp.assert_image_close(got.cpu().numpy(), ref[nm], hp.IMG_TOL, st=st, tag=tag + "/" + nm)
        gref = orc.backward(st, dLc[k], dLd[k], cam["projmatrix_raw"])
        em = gref["error_model"] = orc.error_model(st, dLc[k], dLd[k], hp.BORDER_REL, hp.BORDER_REL_T)
        sums12 = bc.view_sums(k).cpu().numpy()
        gv = hp.view_grads_from_sums(sums12, g["tau"][k].cpu().numpy(), g["tau_all"][k].cpu().numpy())
        np.testing.assert_array_equal(gv[0], g["mean2D"][k].cpu().numpy())  # (the exported sums ARE what the chain consumed)
        hp.assert_grads_close(gv, gref, tag, st=st, projmatrix_raw=cam["projmatrix_raw"])
        # the per-view fp64 chains on the device's sums, and the per-row bounds of layer (B), for the window sums below
        dsums = (gv[0], gv[10], gv[1], gv[11])
        tr, sens, noise32, _ = hp.chain_sensitivity(st, dsums, cam["projmatrix_raw"])
        for n in chain_names:
            tt = tr[n].reshape(P, -1)
            scale = np.maximum(np.abs(tt).max(axis=1), hp.CHAIN_FLOOR * np.abs(tt).max())
            truth[n] = truth[n] + tt
            allowed[n] = allowed[n] + np.maximum(np.maximum(hp.CHAIN_ROW_TOL * scale, hp.CHAIN_COND_K * sens[n]), hp.CHAIN_K * noise32[n])
        mass, cond, flip = (em[kk].astype(np.float64)[:, 5] for kk in ("term_mass", "cond_slack", "flip_budget"))
        op_want = op_want + gref["dL_dopacity"].astype(np.float64).reshape(P)
        op_bound = op_bound + hp.MASS_TOL * mass + hp.COND_K * cond + hp.FLIP_K * flip + 1e-9 * np.abs(gref["dL_dopacity"]).max() + 1e-37
    worst = {}
    for n, key in (("dL_dmean3D", "mean3D"), ("dL_dsh", "sh"), ("dL_dscale", "scale"), ("dL_drot", "rot")):
        err = np.abs(f64(g[key]).reshape(P, -1) - truth[n]).max(axis=1)
        # (+ the K - 1 fp32 additions of the sum itself)
        bound = allowed[n] + K * 2.0 ** -23 * np.abs(truth[n]).max(axis=1)
        i = int(np.argmax(err / bound))
        worst[n] = float((err / bound).max())
        assert err[i] <= bound[i], ("bench_window/sum", n, "row %d: error %.3e, sum of the per-view bounds %.3e" % (i, err[i], bound[i]))
    err = np.abs(f64(g["opacity"]).reshape(P) - op_want)
    worst["dL_dopacity"] = float((err / op_bound).max())
    assert (err <= op_bound).all(), ("bench_window/sum", "dL_dopacity", float((err / op_bound).max()))
    hp._errlog("bench_window/sums_err_over_bound", **worst)
