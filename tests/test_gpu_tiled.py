"""GPU parity of the tiled rasteriser (forward + analytical-Jacobian backward) against the CPU
oracle, through the C ABI.  Tolerances (tests/helpers.py): integers and indices exact; n_contrib and
images may differ only at pixels whose oracle walk meets a cut-off borderline contributor (each such
pixel is verified), all other pixels within IMG_TOL = 2e-5 of the image's max; every gradient within
GRAD_TOL = 1e-4 of its tensor's max and every row within ROW_TOL = 1e-3 of its own magnitude (the oracle
sums per-Gaussian contributions in fp64, the kernels in fp32 with a different association order, and
use v_exp_f32 / v_rcp_f32 where the oracle uses libm expf and a true division)."""
import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu

IMG_TOL = hp.IMG_TOL
GRAD_TOL = hp.GRAD_TOL


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("name", list(hp.SCENES))
@pytest.mark.parametrize("precomp", [False, True])
def test_forward_and_backward_parity(torch_cuda, name, precomp):
    from gsaj import rasterizer as C

    cam, sc, deg = hp.make(name)
    bg = (0.1, 0.2, 0.3)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg, precomp=precomp)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    assert R == ref["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["tiles_touched"], st["tiles_touched"])
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    vis = ref["radii"] > 0
    assert hp.rel_err(dbg["means2D"][vis], st["means2D"][vis]) < 1e-6
    assert hp.rel_err(dbg["depths"][vis], st["depths"][vis]) < 1e-6
    assert hp.rel_err(dbg["conic_opacity"][vis], st["conic_opacity"][vis]) < 1e-5
    if not precomp:
        assert hp.rel_err(dbg["rgb"][vis], st["rgb"][vis]) < 1e-5
        np.testing.assert_array_equal(dbg["clamped"][vis], st["clamped"][vis])
        assert hp.rel_err(dbg["cov3D"][vis], st["cov3D"][vis]) < 1e-6
    # integer image-space outputs: exact except at pixels with a cut-off borderline contributor (v_exp_f32 vs expf)
    tag = "%s/%s" % (name, "precomp" if precomp else "sh")
    hp.assert_counts_close(dbg["n_contrib"], st["n_contrib"], st, tag=tag)
    hp.assert_touched_close(n_touched.cpu().numpy(), ref["n_touched"], st, tag=name + "/n_touched")
    for nm, got, want in (("color", color, ref["color"]), ("depth", depth, ref["depth"]), ("opacity", opacity, ref["opacity"]),
                          ("final_T", dbg["final_T"], st["final_T"])):
        got = got.cpu().numpy() if hasattr(got, "cpu") else got
        hp.assert_image_close(got.reshape(want.shape), want, IMG_TOL, st=st, tag=tag + "/" + nm)

    dLc, dLd = hp.seeds(cam, seed=1)
    g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, tag)
    dtau, dtau_sum = g[8].cpu().numpy(), g[9].cpu().numpy()
    assert hp.rel_err(dtau.astype(np.float64).sum(0), dtau_sum) < 1e-5


def test_bitwise_reproducible(torch_cuda):
    """No float atomics anywhere: two runs give identical bits."""
    cam, sc, deg = hp.make("p6000_640x480_sh1")
    (_, _), kw = hp.oracle_forward(cam, sc, deg)
    dLc, dLd = hp.seeds(cam, seed=2)
    runs = []
    for _ in range(2):
        out, args = hp.gpu_forward(cam, sc, deg, kw=kw)
        g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
        runs.append([out[1].cpu().numpy(), out[8].cpu().numpy()] + [x.cpu().numpy() for x in g if x is not None])
    for a, b in zip(*runs):
        assert np.array_equal(a, b)


def test_all_culled_and_errors(torch_cuda):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    cam, sc, deg = hp.make("p300_behind_64x48")
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
    # move everything behind the camera: R = 0, images = background
    c2w = np.linalg.inv(cam["w2c"])
    behind = (np.array([0, 0, -5.0, 1.0]) @ c2w.T)[:3].astype(np.float32)
    means = np.tile(behind, (50, 1)) + np.random.default_rng(0).normal(scale=0.1, size=(50, 3)).astype(np.float32)
    settings = GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=t(np.array([0.5, 0.25, 0.125], np.float32)), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
        projmatrix=t(cam["projmatrix"]), projmatrix_raw=t(cam["projmatrix_raw"]), sh_degree=0,
        campos=t(cam["campos"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    m3 = t(means).requires_grad_(True)
    op = t(np.full((50, 1), 0.5, np.float32))
    col = t(np.full((50, 3), 0.5, np.float32))
    sc_ = t(np.full((50, 3), 0.05, np.float32))
    rot = t(np.tile(np.array([1, 0, 0, 0], np.float32), (50, 1)))
    color, radii, depth, opacity, n_touched = rast(m3, torch.zeros_like(m3), op, colors_precomp=col, scales=sc_, rotations=rot)
    assert int(radii.max()) == 0 and int(n_touched.max()) == 0
    assert torch.allclose(color[:, 0, 0], settings.bg) and float(opacity.abs().max()) == 0.0
    color.sum().backward()
    assert float(m3.grad.abs().max()) == 0.0
    assert not bool(rast.markVisible(m3).any())
    # prefiltered + culled point: the reference traps; here a loud error
    with pytest.raises(Exception):
        GaussianRasterizer(settings._replace(prefiltered=True))(m3, torch.zeros_like(m3), op, colors_precomp=col, scales=sc_, rotations=rot)
    # argument-combination errors carry the reference's messages
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(m3, torch.zeros_like(m3), op, scales=sc_, rotations=rot)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(m3, torch.zeros_like(m3), op, colors_precomp=col)
    with pytest.raises(RuntimeError, match="num_points, 3"):
        rast(t(np.zeros((5, 2), np.float32)), torch.zeros(5, 2, device=dev), op[:5], colors_precomp=col[:5], scales=sc_[:5], rotations=rot[:5])


def test_render_api_autograd_and_pose_update(torch_cuda):
    """gaussian_renderer.render() + loss.backward() + pose_utils.update_pose, as slam_frontend.tracking does."""
    import torch
    from gaussian_splatting.gaussian_renderer import render
    from gaussian_splatting.scene.gaussian_model import GaussianModel
    from utils.camera_utils import Camera
    from utils.pose_utils import update_pose
    from oracle import oracle as orc

    cam, sc, deg = hp.make("p2000_160x120")
    dev = "cuda:0"
    model = GaussianModel.from_activated(sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["shs"],
                                         sh_degree=3, device=dev)
    view = Camera.from_synthetic(cam, device=dev)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False

    bg = torch.tensor([0.0, 0.0, 0.0], device=dev)
    pkg = render(view, model, Pipe, bg)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth", "opacity", "n_touched"}
    gen = torch.Generator(device="cpu").manual_seed(0)
    wc = torch.randn(3, cam["H"], cam["W"], generator=gen).to(dev)
    wd = torch.randn(1, cam["H"], cam["W"], generator=gen).to(dev)
    loss = (pkg["render"] * wc).sum() / wc.numel() + (pkg["depth"] * wd).sum() / wd.numel()
    loss.backward()
    # oracle with the same (torch-computed) matrices and activated parameters
    f = lambda x: x.detach().cpu().numpy()  # noqa: E731
    ref, st = orc.forward(f(model.get_xyz), f(model.get_opacity), f(view.world_view_transform), f(view.full_proj_transform),
                          f(view.camera_center), cam["tanfovx"], cam["tanfovy"], cam["W"], cam["H"], np.zeros(3, np.float32),
                          shs=f(model.get_features), scales=f(model.get_scaling), rotations=f(model.get_rotation), sh_degree=3)
    hp.assert_image_close(f(pkg["render"]), ref["color"], IMG_TOL, st=st)
    g = orc.backward(st, f(wc) / wc.numel(), f(wd) / wd.numel(), f(view.projection_matrix))
    tau = g["dL_dtau_sum"]
    assert hp.rel_err(f(view.cam_trans_delta.grad), tau[:3]) < GRAD_TOL
    assert hp.rel_err(f(view.cam_rot_delta.grad), tau[3:]) < GRAD_TOL
    assert hp.rel_err(f(model._xyz.grad), g["dL_dmean3D"]) < GRAD_TOL
    assert hp.rel_err(f(pkg["viewspace_points"].grad), g["dL_dmean2D"]) < GRAD_TOL
    assert model._features_dc.grad is not None and model._scaling.grad is not None and model._rotation.grad is not None
    # one gradient step on the pose and the left-multiplicative update
    with torch.no_grad():
        view.cam_trans_delta -= 1e-3 * view.cam_trans_delta.grad
        view.cam_rot_delta -= 1e-3 * view.cam_rot_delta.grad
    R_before = view.R.clone()
    converged = update_pose(view)
    assert converged.dtype == torch.bool and float(view.cam_rot_delta.abs().max()) == 0.0
    assert not torch.equal(R_before, view.R)
    # masked render keeps full-length bookkeeping outputs
    mask = torch.zeros(sc["means3D"].shape[0], dtype=torch.bool, device=dev)
    mask[::2] = True
    pkg2 = render(view, model, Pipe, bg, mask=mask)
    assert pkg2["radii"].shape[0] == mask.shape[0] and int(pkg2["radii"][~mask].max()) == 0


def test_dL_dtau_matches_finite_differences(torch_cuda):
    """The analytical pose Jacobian against central differences of the HIP forward itself under
    W2C <- Exp(delta) W2C (the method of VerifyJacobian.ipynb, evaluated at delta = 0), in the
    smooth regime of tests/test_oracle_fd.py (one tile, wide Gaussians, no cut-off active,
    SH degree 0, cx = W/2)."""
    import torch
    from gaussian_splatting.gaussian_renderer import render
    from gaussian_splatting.scene.gaussian_model import GaussianModel
    from utils.camera_utils import Camera
    from utils.pose_utils import SE3_exp
    from gsaj import synthetic as syn

    W = H = 16
    kw = dict(W=W, H=H, fx=16.0, fy=16.0, cx=W / 2, cy=H / 2)
    cam = syn.fixture_camera(noisy=True, orthonormal=True, **kw)
    sc = syn.make_scene(5, 3, cam, z_range=(2.0, 3.0), log_scale_range=(np.log(2.0), np.log(3.0)), sh_coeffs=16,
                        opacity_range=(0.25, 0.35), margin=-0.3)
    dev = "cuda:0"
    model = GaussianModel.from_activated(sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["shs"],
                                         sh_degree=3, active_sh_degree=0, device=dev)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False

    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    rng = np.random.default_rng(0)
    wc = torch.tensor(rng.normal(size=(3, H, W)), device=dev)
    wd = torch.tensor(rng.normal(size=(1, H, W)), device=dev)

    def loss_at(tau64):
        T = SE3_exp(torch.tensor(tau64, dtype=torch.float64)) @ torch.tensor(cam["w2c"], dtype=torch.float64)
        view = Camera.from_synthetic(syn.make_camera(T.numpy(), **kw), device=dev)
        pkg = render(view, model, Pipe, bg)
        return view, (pkg["render"].double() * wc).sum() + (pkg["depth"].double() * wd).sum()

    view, loss = loss_at(np.zeros(6))
    loss.backward()
    ana = np.concatenate([view.cam_trans_delta.grad.cpu().numpy(), view.cam_rot_delta.grad.cpu().numpy()])
    num = np.zeros(6)
    eps = 2e-3
    for k in range(6):
        d = np.zeros(6)
        d[k] = eps
        with torch.no_grad():
            num[k] = (float(loss_at(d)[1]) - float(loss_at(-d)[1])) / (2 * eps)
    assert np.abs(ana - num).max() < 1e-2 * np.abs(num).max(), (ana, num)


@pytest.mark.parametrize("mode", ["forced", "long_lists_in_lds", "oversized_tile"])
def test_long_tile_lists(torch_cuda, mode):
    """The tile sort has two paths: one LDS pass (lists up to the capacity the launch was sized for, at most 16384 entries = 128 KB
    of LDS) and LDS-sized chunks + merge passes inside the tile's workgroup for longer lists (the reference's global radix sort
    takes any length, rasterizer_impl.cu:353-368).  Both must give the oracle's order bit for bit -- "forced": every list longer
    than 128 entries through chunks of 128; lists of ~6000 entries (more dynamic LDS than the 64 KB default limit); lists
    beyond the LDS capacity."""
    from gsaj import rasterizer as C
    from gsaj import synthetic as syn
    from oracle import oracle as orc

    if mode == "forced":
        cam, sc, deg = hp.make("p2000_160x120")
    else:  # 6000 / 18000 Gaussians on a 32x32 image: every tile list has ~6000 (LDS, > 64 KB) / > 16384 (chunks + merge) entries
        n = 6000 if mode == "long_lists_in_lds" else 18000
        cam = hp.small_camera(32, 32, f=30.0, orthonormal=True)
        sc = syn.make_scene(n, 9, cam, z_range=(1.0, 3.0), log_scale_range=(np.log(0.2), np.log(0.5)), sh_coeffs=1,
                            opacity_range=(0.01, 0.05) if n == 6000 else (0.004, 0.02), margin=-0.1)
        deg = 0
    (ref, st), kw = hp.oracle_forward(cam, sc, deg)
    C.FORCE_CHUNKED_SORT = mode == "forced"
    try:
        out, args = hp.gpu_forward(cam, sc, deg, kw=kw)
    finally:
        C.FORCE_CHUNKED_SORT = False
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    longest = (st["ranges"][:, 1] - st["ranges"][:, 0]).max()
    if mode == "long_lists_in_lds":
        assert 8192 >= longest > 4096 or 16384 >= longest > 4096
    if mode == "oversized_tile":
        assert longest > 16384
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(sc["means3D"].shape[0], R, cam["W"], cam["H"], geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    hp.assert_image_close(color.cpu().numpy(), ref["color"], IMG_TOL, st=st)
    dLc, dLd = hp.seeds(cam, seed=3)
    g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, "long_lists/" + mode)


def test_async_forward_matches_sync_and_reports_overflow(torch_cuda):
    """FrameContext.forward(sync=False): no host round trip; same bits as the synchronous path; a frame
    that does not fit the arena is aborted on the device and reported by status()."""
    import torch
    from gsaj import _lib
    from gsaj.rasterizer import FrameContext

    cam, sc, deg = hp.make("p6000_640x480_sh1")
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    args = dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), opacities=t(sc["opacities"]),
                viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]),
                tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=deg, shs=t(sc["shs"]), scales=t(sc["scales"]),
                rotations=t(sc["rotations"]))
    dLc, dLd = hp.seeds(cam, seed=4)
    bargs = dict(bg=args["bg"], means3D=args["means3D"], viewmatrix=args["viewmatrix"], projmatrix=args["projmatrix"],
                 projmatrix_raw=t(cam["projmatrix_raw"]), campos=args["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                 dL_dcolor=t(dLc), dL_ddepth=t(dLd), sh_degree=deg, shs=args["shs"], scales=args["scales"],
                 rotations=args["rotations"])
    ctx = FrameContext(P, cam["W"], cam["H"], M, dev)
    ctx.forward(**args, sync=True)
    g = ctx.backward(**bargs)
    ref = [ctx.color.clone(), ctx.depth.clone(), ctx.n_touched.clone(), ctx.bucket.clone(), g["tau_sum"].clone()]
    R_true = ctx.R
    ctx.forward(**args, sync=False)
    g = ctx.backward(**bargs)
    assert ctx.status()[0] == R_true
    for a, b in zip(ref, [ctx.color, ctx.depth, ctx.n_touched, ctx.bucket, g["tau_sum"]]):
        assert torch.equal(a, b)
    # shrink the arena below R: the frame must abort on the device and be reported
    ctx.capacity = R_true // 2
    ctx.forward(**args, sync=False)
    with pytest.raises(_lib.GsajError, match="too small|aborted"):
        ctx.status()
    # a tile list longer than the LDS sort was sized for is NOT an abort: it is sorted in chunks + merge passes, same bits
    ctx2 = FrameContext(P, cam["W"], cam["H"], M, dev)
    ctx2.forward(**args, sync=True)
    longest = ctx2.status()[1]
    assert ctx2.tile_list_capacity >= longest
    ctx2.forward(**args, sync=False)
    assert torch.equal(ctx2.color, ref[0])
    for cap in (max(1, longest // 2), 1):
        ctx2.tile_list_capacity = cap
        ctx2.forward(**args, sync=False)
        g2 = ctx2.backward(**bargs)
        assert ctx2.status()[0] == R_true
        for a, b in zip(ref, [ctx2.color, ctx2.depth, ctx2.n_touched, ctx2.bucket, g2["tau_sum"]]):
            assert torch.equal(a, b)


def test_mark_visible_matches_oracle(torch_cuda):
    """GaussianRasterizer.markVisible (in_frustum: view-space z > 0.2, auxiliary.h:139-164) against the oracle, on points
    in front of, behind and right at the near threshold of the camera."""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import oracle as orc

    cam = hp.small_camera(160, 120, orthonormal=True)
    rng = np.random.default_rng(3)
    w2c = np.asarray(cam["w2c"], np.float64)
    pc = np.concatenate([rng.uniform(-2, 2, (4000, 2)), rng.uniform(-1.0, 3.0, (4000, 1))], axis=1)   # camera-frame points
    pc[:50, 2] = 0.2 + rng.uniform(-1e-3, 1e-3, 50)                                                       # around the threshold
    pw = ((np.linalg.inv(w2c) @ np.concatenate([pc, np.ones((4000, 1))], axis=1).T).T[:, :3]).astype(np.float32)
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    st = GaussianRasterizationSettings(image_height=120, image_width=160, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
                                       projmatrix=t(cam["projmatrix"]), projmatrix_raw=t(cam["projmatrix_raw"]), sh_degree=0,
                                       campos=t(cam["campos"]), prefiltered=False, debug=False)
    got = GaussianRasterizer(st).markVisible(t(pw)).cpu().numpy().astype(bool)
    want = orc.mark_visible(pw, cam["viewmatrix"]).astype(bool)
    assert got.shape == want.shape and 500 < want.sum() < 3500
    np.testing.assert_array_equal(got, want)


def test_pose_only_backward_gives_the_same_tau(torch_cuda):
    """FrameContext.backward(pose_only=True): dL/dtau bit-identical to the full backward, per-Gaussian buffers untouched."""
    import torch
    from gsaj.rasterizer import FrameContext

    cam, sc, deg = hp.make("p6000_640x480_sh1")
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    args = dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), viewmatrix=t(cam["viewmatrix"]),
                projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=deg,
                shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    dLc, dLd = hp.seeds(cam, seed=8)
    bargs = dict(bg=args["bg"], means3D=args["means3D"], viewmatrix=args["viewmatrix"], projmatrix=args["projmatrix"],
                 projmatrix_raw=t(cam["projmatrix_raw"]), campos=args["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                 dL_dcolor=t(dLc), dL_ddepth=t(dLd), sh_degree=deg, shs=args["shs"], scales=args["scales"], rotations=args["rotations"])
    ctx = FrameContext(P, cam["W"], cam["H"], M, dev, per_gaussian_tau=True)
    ctx.forward(**args)
    g = ctx.backward(**bargs)
    tau_full, tau_pg = g["tau_sum"].clone(), g["tau"].clone()
    ctx.bucket.fill_(123.0)
    g = ctx.backward(**bargs, pose_only=True)
    assert torch.equal(g["tau_sum"], tau_full) and torch.equal(g["tau"], tau_pg)
    assert bool((ctx.bucket == 123.0).all())
