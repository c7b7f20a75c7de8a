"""GPU: randomized parity sweep of the tiled path against the oracle -- odd image sizes, Gaussian counts
that are not multiples of the workgroup sizes, screen-filling and sub-pixel Gaussians, opacity extremes,
points around the near plane, similarity-transform cameras -- plus size-independent properties at the
full cfg2 size (50 000 Gaussians, 640x480), where the oracle is only used for integer structure."""
import math

import numpy as np
import pytest

import helpers as hp
from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu

CASES_LARGE_IMAGE = [(3000, 2064, 2000, 9)]  # 129 x 125 = 16 125 tiles > LDS_TILES_MAX (8192): global-atomic binning path


@pytest.mark.parametrize("case", CASES_LARGE_IMAGE, ids=["P3000_2064x2000"])
def test_more_tiles_than_the_lds_histogram_holds(case):
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    P, W, H, seed = case
    cam = hp.small_camera(W, H, f=0.9 * W, orthonormal=True)
    sc = syn.make_scene(P, seed, cam, z_range=(1.0, 4.0), log_scale_range=(math.log(0.004), math.log(0.05)), sh_coeffs=4, margin=0.1)
    (ref, st), kw = hp.oracle_forward(cam, sc, 1)
    out, args = hp.gpu_forward(cam, sc, 1, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"]
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    hp.assert_image_close(color.cpu().numpy(), ref["color"], 2e-4)
    dLc, dLd = hp.seeds(cam, seed=seed)
    gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
    g = hp.gpu_backward(cam, 1, out, args, dLc, dLd)
    assert hp.rel_err(g[3].cpu().numpy().reshape(gref["dL_dmean3D"].shape), gref["dL_dmean3D"]) < 3e-3
    assert hp.rel_err(g[9].cpu().numpy(), gref["dL_dtau_sum"]) < 3e-3


CASES = [
    # P, W, H, seed, deg, coeffs, z_range, log_scale_range, opacity_range, orthonormal
    (1, 33, 17, 1, 0, 1, (1.0, 1.5), (math.log(0.05), math.log(0.1)), (0.5, 0.9), True),
    (63, 47, 31, 2, 1, 4, (0.5, 2.0), (math.log(0.02), math.log(0.3)), (0.01, 0.99), False),
    (65, 16, 16, 3, 3, 16, (1.0, 2.0), (math.log(0.5), math.log(2.0)), (0.2, 0.6), True),      # screen-filling
    (257, 129, 65, 4, 2, 9, (0.15, 1.0), (math.log(0.002), math.log(0.02)), (0.3, 0.95), False),  # near plane, tiny
    (1000, 200, 150, 5, 3, 16, (1.0, 8.0), (math.log(0.003), math.log(0.6)), (0.004, 1.0), True),  # wide mix
    (4097, 96, 96, 6, 0, 1, (1.0, 3.0), (math.log(0.01), math.log(0.05)), (0.9, 0.999), True),   # near-opaque
]


@pytest.mark.parametrize("case", CASES, ids=[f"P{c[0]}_{c[1]}x{c[2]}" for c in CASES])
def test_random_scene_parity(case):
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    P, W, H, seed, deg, coeffs, zr, ls, orng, ortho = case
    cam = hp.small_camera(W, H, f=0.8 * W, orthonormal=ortho)
    sc = syn.make_scene(P, seed, cam, z_range=zr, log_scale_range=ls, opacity_range=orng, sh_coeffs=coeffs, margin=0.2)
    bg = (0.3, 0.1, 0.7)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    assert (dbg["n_contrib"].astype(np.int64) != st["n_contrib"].astype(np.int64)).mean() <= 2e-4
    hp.assert_image_close(color.cpu().numpy(), ref["color"], 2e-4)
    hp.assert_image_close(depth.cpu().numpy(), ref["depth"], 2e-4)
    dLc, dLd = hp.seeds(cam, seed=seed)
    gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
    g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dtau"]
    for nm, got in zip(names, g[:9]):
        want = gref[nm]
        if np.abs(want).max() == 0:
            assert float(got.abs().max()) == 0, nm
            continue
        assert hp.rel_err(got.cpu().numpy().reshape(want.shape), want) < 3e-3, (nm, hp.rel_err(got.cpu().numpy().reshape(want.shape), want))
    assert hp.rel_err(g[9].cpu().numpy(), gref["dL_dtau_sum"]) < 3e-3


def test_full_size_properties_cfg2():
    """50 000 Gaussians at 640x480 (the benchmark workload): structure against the oracle, plus
    properties that need no oracle -- sorted lists, linearity of the backward in the pixel seeds,
    idempotence, opacity + final_T = 1, counts."""
    import torch
    from gsaj import rasterizer as C
    from gsaj.rasterizer import FrameContext

    cam, sc = syn.config_scene("cfg2")
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], 16
    (ref, st), kw = hp.oracle_forward(cam, sc, 3)
    out, args = hp.gpu_forward(cam, sc, 3, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"] == int(st["tiles_touched"].sum())
    dbg = {k: v for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    np.testing.assert_array_equal(dbg["point_list"].cpu().numpy().astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"].cpu().numpy(), st["ranges"])
    # every tile list is sorted by (depth, id)
    pl = dbg["point_list"].long()
    d = dbg["depths"][pl]
    rg = dbg["ranges"].long()
    tile_of = torch.repeat_interleave(torch.arange(rg.shape[0], device=pl.device), rg[:, 1] - rg[:, 0])
    same = tile_of[1:] == tile_of[:-1]
    assert bool(((d[1:] > d[:-1]) | ((d[1:] == d[:-1]) & (pl[1:] > pl[:-1])) | ~same).all())
    assert torch.allclose(opacity[0] + dbg["final_T"], torch.ones_like(dbg["final_T"]), atol=1e-6)
    assert int((dbg["n_contrib"].long() > (rg[:, 1] - rg[:, 0]).view(H // 16, W // 16).repeat_interleave(16, 0).repeat_interleave(16, 1)).sum()) == 0
    assert (dbg["n_contrib"].cpu().numpy().astype(np.int64) != st["n_contrib"].astype(np.int64)).mean() <= 1e-4
    hp.assert_image_close(color.cpu().numpy(), ref["color"], 2e-4)
    # linearity of the backward in the seeds + idempotence (same inputs, same bits)
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    ctx = FrameContext(P, W, H, M, dev)
    fa = dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), viewmatrix=t(cam["viewmatrix"]),
              projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3,
              shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    ctx.forward(**fa)
    s1c, s1d = hp.seeds(cam, seed=11)
    s2c, s2d = hp.seeds(cam, seed=12)

    def bwd(c, dd):
        g = ctx.backward(bg=fa["bg"], means3D=fa["means3D"], viewmatrix=fa["viewmatrix"], projmatrix=fa["projmatrix"],
                         projmatrix_raw=t(cam["projmatrix_raw"]), campos=fa["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                         dL_dcolor=t(c), dL_ddepth=t(dd), sh_degree=3, shs=fa["shs"], scales=fa["scales"], rotations=fa["rotations"])
        return ctx.bucket.clone(), g["tau_sum"].clone()

    b1, t1 = bwd(s1c, s1d)
    b2, t2 = bwd(s2c, s2d)
    b3, t3 = bwd(2.0 * s1c - 0.5 * s2c, 2.0 * s1d - 0.5 * s2d)
    b1b, t1b = bwd(s1c, s1d)
    assert torch.equal(b1, b1b) and torch.equal(t1, t1b)
    scale = float(b3.abs().max())
    assert float((b3 - (2.0 * b1 - 0.5 * b2)).abs().max()) < 2e-4 * scale
    assert float((t3 - (2.0 * t1 - 0.5 * t2)).abs().max()) < 2e-4 * float(t3.abs().max())


@pytest.mark.parametrize("wl", ["cfg3", "cfg5"])
def test_large_config_properties_without_oracle(wl):
    """cfg3 (300 000 Gaussians, 1200x680) and cfg5 (10^6 Gaussians, 1280x720): too large for the CPU oracle inside a test,
    so only properties that need no oracle -- instance count = sum of tiles touched, every tile list sorted by (depth, id),
    opacity + final_T = 1, n_contrib within the list, bit-identical repeat, backward linear in the pixel seeds."""
    import torch
    from gsaj import rasterizer as C
    from gsaj.rasterizer import FrameContext

    cam, sc = syn.config_scene(wl)
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    deg = int(round(M ** 0.5)) - 1
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    ctx = FrameContext(P, W, H, M, dev)
    fa = dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), viewmatrix=t(cam["viewmatrix"]),
              projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=deg,
              shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    ctx.forward(**fa)
    R, longest = ctx.status()
    dbg = C.debug_export(P, R, W, H, ctx.geom, ctx.binning, ctx.img)
    assert R == int(dbg["tiles_touched"].long().sum()) and longest <= 4096
    pl, rg = dbg["point_list"].long(), dbg["ranges"].long()
    d = dbg["depths"][pl]
    tile_of = torch.repeat_interleave(torch.arange(rg.shape[0], device=dev), rg[:, 1] - rg[:, 0])
    same = tile_of[1:] == tile_of[:-1]
    assert bool(((d[1:] > d[:-1]) | ((d[1:] == d[:-1]) & (pl[1:] > pl[:-1])) | ~same).all())
    assert int((rg[:, 1] - rg[:, 0]).max()) == longest
    assert torch.allclose(ctx.opacity[0] + dbg["final_T"], torch.ones_like(dbg["final_T"]), atol=1e-6)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    per_pixel_len = (rg[:, 1] - rg[:, 0]).view(gy, gx).repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    assert int((dbg["n_contrib"].long() > per_pixel_len).sum()) == 0
    color1 = ctx.color.clone()
    s1c, s1d = hp.seeds(cam, seed=21)
    s2c, s2d = hp.seeds(cam, seed=22)

    def bwd(c, dd):
        g = ctx.backward(bg=fa["bg"], means3D=fa["means3D"], viewmatrix=fa["viewmatrix"], projmatrix=fa["projmatrix"],
                         projmatrix_raw=t(cam["projmatrix_raw"]), campos=fa["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                         dL_dcolor=t(c), dL_ddepth=t(dd), sh_degree=deg, shs=fa["shs"], scales=fa["scales"], rotations=fa["rotations"])
        return ctx.bucket.clone(), g["tau_sum"].clone()

    b1, t1 = bwd(s1c, s1d)
    b2, t2 = bwd(s2c, s2d)
    b3, t3 = bwd(s1c + 3.0 * s2c, s1d + 3.0 * s2d)
    ctx.forward(**fa)
    assert torch.equal(ctx.color, color1)
    b1b, t1b = bwd(s1c, s1d)
    assert torch.equal(b1, b1b) and torch.equal(t1, t1b)
    assert float((b3 - (b1 + 3.0 * b2)).abs().max()) < 3e-4 * float(b3.abs().max())
    assert float((t3 - (t1 + 3.0 * t2)).abs().max()) < 3e-4 * float(t3.abs().max())
