"""GPU: gsaj.tracking.DeviceTracker -- the tracking loop of a frame kept on the device, one iteration captured
into a hipGraph and replayed.  The replayed loop must land on the SAME BITS as the eager sequence of C-ABI calls (same kernels,
same launch parameters, same order), follow the whole-frame loop when the frame is sharded into tile bands, survive a new frame
without re-capturing, and recover when a frame outgrows the arena."""
import math

import numpy as np
import pytest

from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu


def _setup(P=3000, W=160, H=120, seed=11):
    import torch
    from gsaj.rasterizer import FrameContext

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    f = 0.875 * W
    cam_gt = syn.fixture_camera(noisy=False, orthonormal=True, W=W, H=H, fx=f, fy=f, cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    cam0 = syn.fixture_camera(noisy=True, orthonormal=True, W=W, H=H, fx=f, fy=f, cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    sc = syn.make_scene(P, seed, cam_gt, z_range=(1.0, 4.0), log_scale_range=(math.log(0.02), math.log(0.1)))
    M = sc["shs"].shape[1]
    g = dict(means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]),
             sh_degree=3)
    bg = torch.zeros(3, device=dev)
    gt = FrameContext(P, W, H, M, dev)
    gt.forward(bg, g["means3D"], g["opacities"], t(cam_gt["viewmatrix"]), t(cam_gt["projmatrix"]), t(cam_gt["campos"]), cam_gt["tanfovx"],
               cam_gt["tanfovy"], sh_degree=3, shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    w2c0 = np.ascontiguousarray(cam0["viewmatrix"].T)
    return dev, cam0, g, bg, gt.color.clone(), gt.depth[0].clone(), w2c0, M, t(cam0["projmatrix_raw"])


def _tracker(dev, cam, g, bg, w2c, M, praw, **kw):
    from gsaj.tracking import DeviceTracker

    P = g["means3D"].shape[0]
    return DeviceTracker(P, cam["W"], cam["H"], M, dev, w2c, praw, cam["tanfovx"], cam["tanfovy"], bg, alpha=0.9, **g, **kw)


def test_graph_replay_equals_eager_bit_for_bit_and_converges():
    import torch

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    out = {}
    for use_graph in (False, True):
        tr = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=use_graph)
        tr.set_frame(gt_c, gt_d)
        losses = []
        for _ in range(4):
            assert tr.iterate(5) == 5
            losses.append(float(tr.loss_terms[0]))
        out[use_graph] = (tr.w2c.clone(), losses, tr.pose.state.clone())
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][2], out[False][2])
    assert out[True][1] == out[False][1]
    assert out[True][1][-1] < 0.75 * out[True][1][0]  # 20 iterations (lr 1e-3 / 3e-3) towards the ground-truth pose: 0.133 -> 0.082


def test_graph_replay_survives_host_synchronisation_between_replays():
    """status() (a blocking read-back) or any other host synchronisation between two replays must not disturb the next one
    (with hipMemsetAsync inside the captured forward it did on ROCm 7.2: the histogram was no longer cleared)."""
    import torch

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    a = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=True)
    b = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=False)
    for tr in (a, b):
        tr.set_frame(gt_c, gt_d)
    for _ in range(6):
        a.iterate(1)  # ends with status(): two blocking read-backs between consecutive replays
        b.iterate(1)
        torch.cuda.synchronize()
        assert torch.equal(a.w2c, b.w2c)


def test_new_frame_reuses_the_graph_and_reset_is_in_place():
    import torch

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    tr = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=True)
    tr.set_frame(gt_c, gt_d)
    tr.iterate(6)
    first = tr.w2c.clone()
    graphs = tr._graphs
    assert graphs is not None
    tr.set_frame(gt_c, gt_d, w2c=w2c0)  # same frame again from the same start: same result, same graph objects
    tr.iterate(6)
    assert tr._graphs is graphs
    assert torch.equal(tr.w2c, first)
    with pytest.raises(Exception, match="every frame"):
        tr.set_frame(gt_c, None)


def test_early_exit_on_the_device_converged_flag():
    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    tr = _tracker(dev, cam, g, bg, w2c0, M, praw, converged_threshold=1e3)  # any step counts as converged
    tr.set_frame(gt_c, gt_d)
    assert tr.iterate(50, check_every=4) == 4


def test_tile_band_shares_through_the_tracker_follow_the_whole_frame():
    """Three trackers, one band each, stepped in lock step with their packed terms added by hand (what the all-reduce does):
    the pose follows the single whole-frame tracker."""
    import torch
    from gsaj import tile_band_shard as tbs

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    whole = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=False)
    whole.set_frame(gt_c, gt_d)
    whole.iterate(8)
    parts = [_tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=False, band=b) for b in tbs.uniform_bands(cam["H"], 3)]
    for p in parts:
        p.set_frame(gt_c, gt_d)
    for it in range(8):
        total = torch.zeros(tbs.REDUCED_FLOATS, device=dev)
        for p in parts:
            p._render_and_grads(it == 0)
            total += p.packed  # (dL/dtau and the loss scalars were written straight into it)
        for p in parts:
            p.packed.copy_(total)
            p._step()
    for p in parts:
        assert float((p.w2c - whole.w2c).abs().max()) < 2e-6
    assert torch.equal(parts[0].w2c, parts[1].w2c)  # the replicas take the same step from the same sums


def test_arena_overflow_is_reported_and_the_next_call_recovers():
    import torch
    from gsaj import _lib

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    tr = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=True)
    tr.set_frame(gt_c, gt_d)
    tr.iterate(2)
    good = tr.w2c.clone()
    tr.ctx.capacity = 64  # pretend the arena was sized for a nearly empty view ...
    tr._graphs = None     # ... before the graph was captured (launch arguments are frozen into a captured graph)
    with pytest.raises(_lib.GsajError, match="aborted"):
        tr.iterate(3)
    # the aborted iterations rendered nothing, so their dL/dtau was the last good iteration's: the pose step must have skipped
    # them (gsaj_pose_adam_step's `skip` word = the frame's abort flag) -- pose, Adam moments and step count untouched
    assert torch.equal(tr.w2c, good)
    assert float(tr.pose.state[32]) == 2.0
    tr.set_frame(gt_c, gt_d, w2c=w2c0)
    assert tr.iterate(2) == 2  # re-sized by a synchronous first iteration, graph re-captured if the arena moved
    assert torch.equal(tr.w2c, good)


@pytest.mark.parametrize("monocular,masked", [(False, False), (False, True), (True, False)])
def test_loss_fused_into_the_compositors_gives_the_unfused_gradients_bit_for_bit(monocular, masked):
    """SURVEY 8(f)-1 as written: gsaj_rasterize_forward_loss sums the tracking loss in the forward compositor's epilogue,
    gsaj_rasterize_backward_loss derives the pixel seeds in the reverse compositor's prologue -- no loss kernel, no seed images.
    Against forward -> gsaj_loss_seeds -> backward on the same frame: EVERY gradient output identical bit for bit (the seeds
    come from one shared definition of the per-pixel arithmetic, csrc/loss_terms.h), the five loss scalars equal to rounding
    (they are sums in a different order).  gsaj_loss_seeds itself is pinned to the reference's get_loss_tracking under
    autograd (tests/test_gpu_loss.py, loss_*.npz)."""
    import torch
    from gsaj.losses import MONOCULAR, TRACKING, LossSeeds
    from gsaj.rasterizer import FrameContext

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, W, H = g["means3D"].shape[0], cam["W"], cam["H"]
    view, proj, cp = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"])
    kw = dict(sh_degree=3, shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    ea, eb = torch.tensor([0.07], device=dev), torch.tensor([-0.02], device=dev)
    mask = None
    if masked:
        mask = (torch.rand(H * W, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) > 0.3).to(torch.uint8)
    flags = TRACKING | (MONOCULAR if monocular else 0)
    gtd = None if monocular else gt_d.contiguous()

    a = FrameContext(P, W, H, M, dev, per_gaussian_tau=True)
    a.forward(bg, g["means3D"], g["opacities"], view, proj, cp, cam["tanfovx"], cam["tanfovy"], sync=True, **kw)
    a.forward(bg, g["means3D"], g["opacities"], view, proj, cp, cam["tanfovx"], cam["tanfovy"], sync=False, **kw)
    ls = LossSeeds(W, H, dev)
    L = ls(flags, 0.9, 0.01, a.color, a.depth, a.opacity, gt_c, gtd, mask, ea, eb)
    ga = a.backward(bg, g["means3D"], view, proj, praw, cp, cam["tanfovx"], cam["tanfovy"], L["dL_dcolor"], L["dL_ddepth"], **kw)
    want = {n: x.clone() for n, x in ga.items() if torch.is_tensor(x)}
    want_scalars = ls.scalars.clone()
    assert float(want["tau_sum"].abs().max()) > 0.0

    b = FrameContext(P, W, H, M, dev, per_gaussian_tau=True)
    b.forward(bg, g["means3D"], g["opacities"], view, proj, cp, cam["tanfovx"], cam["tanfovy"], sync=True, **kw)
    scalars = torch.zeros(5, device=dev)
    FL = dict(flags=flags, alpha=0.9, rgb_boundary_threshold=0.01, gt_color=gt_c, gt_depth=gtd, grad_mask=mask, exposure_a=ea, exposure_b=eb,
              scalars=scalars)
    b.forward_loss(FL, bg, g["means3D"], g["opacities"], view, proj, cp, cam["tanfovx"], cam["tanfovy"], **kw)
    gb = b.backward_loss(FL, bg, g["means3D"], view, proj, praw, cp, cam["tanfovx"], cam["tanfovy"], **kw)
    assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth) and torch.equal(a.n_touched, b.n_touched)
    for n, x in want.items():
        assert torch.equal(gb[n], x), "dL/d%s of the fused path differs from the unfused path" % n
    assert float((scalars - want_scalars).abs().max()) <= 2e-6 * float(want_scalars.abs().max()), (scalars, want_scalars)


def test_fused_tracker_follows_the_unfused_tracker_bit_for_bit():
    """DeviceTracker(fused=True), the default: an iteration is forward_loss -> backward_loss -> pose step, with no dL/dpix buffers;
    the pose after 10 iterations equals the unfused tracker's bit for bit (the pose step consumes dL/dtau and dL/d(exposure), both
    functions of the per-pixel seeds... dL/d(exposure) is a SUM of pixel terms in another order: the exposure learning rates are
    set to zero here so that the comparison is exact; with them on, the poses agree to 1e-6)."""
    import torch

    dev, cam, g, bg, gt_c, gt_d, w2c0, M, praw = _setup()
    out = {}
    for fused in (False, True):
        for lr_exp in (0.0, 0.01):
            tr = _tracker(dev, cam, g, bg, w2c0, M, praw, use_graph=False, fused=fused, lr_exposure_a=lr_exp, lr_exposure_b=lr_exp)
            tr.set_frame(gt_c, gt_d)
            assert tr.iterate(10) == 10
            out[fused, lr_exp] = (tr.w2c.clone(), tr.loss_terms.clone())
    assert torch.equal(out[False, 0.0][0], out[True, 0.0][0])
    assert float((out[False, 0.01][0] - out[True, 0.01][0]).abs().max()) < 1e-6
    assert float((out[False, 0.0][1] - out[True, 0.0][1]).abs().max()) <= 2e-6 * float(out[False, 0.0][1].abs().max())
