"""GPU: gsaj.rasterizer.ArenaWatch -- the binning arena of asynchronous frames grows AHEAD of a growing map (no host
synchronisation, no aborted frame); without it the same sequence overflows and is aborted on the device.  The reference sizes its
buffers inside every forward (rasterizer_impl.cu:331-338, one blocking read-back per frame); the asynchronous entry points have no
read-back, so the head room must be kept some other way."""
import math

import numpy as np
import pytest

from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(P=4000, W=160, H=120, seed=5):
    import torch

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    f = 0.875 * W
    cam = syn.fixture_camera(noisy=False, orthonormal=True, W=W, H=H, fx=f, fy=f, cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    sc = syn.make_scene(P, seed, cam, z_range=(1.0, 4.0), log_scale_range=(math.log(0.01), math.log(0.05)))
    g = dict(shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]), sh_degree=3)
    a = (torch.zeros(3, device=dev), t(sc["means3D"]), t(sc["opacities"]))
    return dev, cam, sc["shs"].shape[1], a, g, t


# the map "grows": every frame the Gaussians are 2 % larger (scale_modifier), i.e. ~4 % more (Gaussian, tile) instances
GROWTH = [1.02 ** i for i in range(1, 46)]


def test_single_view_arena_grows_ahead_of_the_frames():
    import torch
    from gsaj import _lib
    from gsaj.rasterizer import FrameContext

    dev, cam, M, a, g, t = _scene()
    cams = (t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), cam["tanfovx"], cam["tanfovy"])
    P = a[1].shape[0]
    res = {}
    for auto in (True, False):
        ctx = FrameContext(P, cam["W"], cam["H"], M, dev)
        ctx.auto_grow = auto
        ctx.watch.every = 1
        R0 = ctx.forward(*a, *cams, **g, sync=True)
        cap0 = ctx.capacity
        for s in GROWTH:
            ctx.forward(*a, *cams, **g, scale_modifier=s, sync=False)
            torch.cuda.synchronize(dev)  # (the test makes "a few frames later" deterministic: the event is complete by the next call)
        if auto:
            R, _ = ctx.status()  # raises if any frame was aborted
            assert R > cap0 > R0, (R, cap0, R0)  # the map outgrew the first arena ...
            assert ctx.watch.grown >= 2 and ctx.capacity >= R  # ... and the arena was re-allocated ahead of it, more than once
            res["color"], res["R"] = ctx.color.clone(), R
        else:
            with pytest.raises(_lib.GsajError, match="aborted"):
                ctx.status()
    # the last asynchronous frame is the frame a fresh synchronous context renders
    ref = FrameContext(P, cam["W"], cam["H"], M, dev)
    assert ref.forward(*a, *cams, **g, scale_modifier=GROWTH[-1], sync=True) == res["R"]
    assert torch.equal(ref.color, res["color"])


def test_batched_window_arena_grows_ahead_and_the_tracker_never_aborts():
    import torch
    from gsaj.rasterizer import BatchContext

    dev, cam, M, a, g, t = _scene(P=3000)
    K = 3
    kc = syn.keyframe_cameras(K, W=cam["W"], H=cam["H"], fx=0.875 * cam["W"], fy=0.875 * cam["W"], cx=cam["W"] / 2 - 0.5, cy=cam["H"] / 2 - 0.5)
    views, projs, cps = (t(np.stack([c[k] for c in kc])) for k in ("viewmatrix", "projmatrix", "campos"))
    bc = BatchContext(K, a[1].shape[0], cam["W"], cam["H"], M, dev)
    bc.watch.every = 1
    st = bc.forward(*a, views, projs, cps, cam["tanfovx"], cam["tanfovy"], **g, sync=True)
    bc._size(int(1.5 * max(r for r, _, _ in st)) + 1024)  # (a window is first sized generously, 12 P per view: start from a tight arena)
    cap0 = bc.capacity
    for s in GROWTH:
        bc.forward(*a, views, projs, cps, cam["tanfovx"], cam["tanfovy"], **g, scale_modifier=s, sync=False)
        torch.cuda.synchronize(dev)
    st = bc.status()
    assert not any(ab for _, _, ab in st) and bc.clear_aborts() == 0
    assert max(r for r, _, _ in st) > cap0 and bc.watch.grown >= 2 and bc.capacity > cap0
