"""The tracking loop of one frame kept on the device ("the tracking loop without host round-trips").

The reference's frontend runs ~100 iterations per frame of render -> get_loss_tracking -> backward -> Adam.step ->
update_pose, reading `converged` back every time (utils/slam_frontend.py:135-193).  Here one iteration is four C-ABI calls that
touch no host state -- asynchronous forward, gsaj_loss_seeds, pose-only backward, gsaj_pose_adam_step (10 launches: one
memset, four forward kernels, the loss, two backward kernels, the pose step) -- with dL/dtau and the loss scalars written side
by side into the buffer the pose step (and, sharded, the all-reduce) reads; `converged` is a device scalar looked at every
`check_every` iterations only.

use_graph=True captures the iteration ONCE into a hipGraph (torch.cuda.CUDAGraph) and replays it.  It is off by default
because it does not pay on this stack (ROCm 7.2, MI355X; tools/device_tracker_bench.py): replay is 0.92x the speed of the eager
launches for a 160x120 / 3000-Gaussian frame and 0.94x at 640x480 / 50 000 -- the iteration is bound by the dependent chain of
~10 short kernels, which a graph replays one after the other just the same, not by their launch cost.  Kept for stacks where it does, and
because it proves the property that matters: an iteration is a pure stream program (the replayed loop gives the eager loop's bits).

With a process group (tile-band sharding, gsaj.tile_band_shard) the collective sits between the backward and the pose step:
[forward(band) .. backward] -> all_reduce(12 floats) -> [pose step]; two graphs when captured.

An asynchronous frame that does not fit its arena is aborted on the device: its kernels return at once and dL/dtau, the loss terms
stay the PREVIOUS iteration's.  The pose step is handed the frame's abort word (gsaj_pose_adam_step `skip`) and then changes
nothing -- no Adam moment, no pose; sharded, the word travels as the 12th float of the all-reduce, so every rank skips the same
iterations and iterate() raises on every rank together.
"""
import torch

from . import _lib, tile_band_shard as tbs
from .losses import MONOCULAR, TRACKING, LossSeeds
from .pose_step import PoseTracker
from .rasterizer import FrameContext


class DeviceTracker:
    def __init__(self, P, W, H, M, device, w2c, projection_matrix, tanfovx, tanfovy, bg, means3D, opacities, sh_degree=0, shs=None,
                 colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, monocular=False, alpha=0.95,
                 rgb_boundary_threshold=0.01, record_bits=32, band=None, group=None, use_graph=False, fused=True, **pose_kw):
        """The Gaussians (device tensors) are those of the current map: the loop reads them, it never writes them.  pose_kw:
        learning rates / betas / eps / converged_threshold of PoseTracker (config["Training"]["lr"] in the reference).
        fused=True (default): the loss is evaluated inside the compositors (gsaj_rasterize_forward_loss / _backward_loss): the
        forward's epilogue sums the loss terms, the reverse compositor derives its pixel seeds itself -- no loss kernel, no seed
        images; the pose follows the unfused loop bit for bit (same per-pixel arithmetic), the loss scalars to rounding."""
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GsajError("DeviceTracker needs a HIP device (there is no CPU path)")
        self.ctx = FrameContext(P, W, H, M, self.dev, has_scales=scales is not None, record_bits=record_bits)
        if band is not None:
            self.ctx.set_tile_band(*band)
        self.loss = LossSeeds(W, H, self.dev)
        self.pose = PoseTracker(w2c, projection_matrix, self.dev, **pose_kw)
        self.praw = self.pose.projection
        self.tanfov = (float(tanfovx), float(tanfovy))
        self.flags = TRACKING | (MONOCULAR if monocular else 0)
        self.alpha, self.thr = float(alpha), float(rgb_boundary_threshold)
        self.bg, self.means, self.opac = bg, means3D, opacities
        self.kw = dict(sh_degree=sh_degree, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                       cov3D_precomp=cov3D_precomp)
        self.group, self.use_graph, self.fused = group, bool(use_graph), bool(fused)
        # dL/dtau and the loss kernel's five scalars land side by side in ONE buffer (the backward and gsaj_loss_seeds are handed
        # views of it): nothing to pack before the all-reduce / the pose step -- three launches fewer per iteration
        self.packed = torch.zeros(tbs.REDUCED_FLOATS, dtype=torch.float32, device=self.dev)
        self.ctx.g["tau_sum"] = self.packed[tbs.TAU]
        self.loss.scalars = self.packed[6:11]
        self._abort_word = self.ctx.abort_flag_tensor()   # int32 [1], inside the image workspace
        self._abort_ptr = self.ctx.abort_flag_ptr()
        self.gt_color = self.gt_depth = self.grad_mask = None
        self._graphs = None
        self.iterations = 0

    # ---- one iteration, in the two halves a collective may sit between -------------------------------------------------
    def _render_and_grads(self, sync):
        p, c = self.pose, self.ctx
        if self.fused and not sync:
            L = dict(flags=self.flags, alpha=self.alpha, rgb_boundary_threshold=self.thr, gt_color=self.gt_color, gt_depth=self.gt_depth,
                     grad_mask=self.grad_mask, exposure_a=p.exposure_a, exposure_b=p.exposure_b, scalars=self.packed[6:11])
            c.forward_loss(L, self.bg, self.means, self.opac, p.viewmatrix, p.projmatrix, p.campos, self.tanfov[0], self.tanfov[1], **self.kw)
            g = c.backward_loss(L, self.bg, self.means, p.viewmatrix, p.projmatrix, self.praw, p.campos, self.tanfov[0], self.tanfov[1],
                                pose_only=True, **self.kw)
            assert g["tau_sum"].data_ptr() == self.packed.data_ptr()
            if self._sharded():
                self.packed[tbs.ABORTED].copy_(self._abort_word)
            return
        c.forward(self.bg, self.means, self.opac, p.viewmatrix, p.projmatrix, p.campos, self.tanfov[0], self.tanfov[1], sync=sync,
                  **self.kw)
        L = self.loss(self.flags, self.alpha, self.thr, c.color, c.depth, c.opacity, self.gt_color, self.gt_depth, self.grad_mask,
                      p.exposure_a, p.exposure_b)
        g = c.backward(self.bg, self.means, p.viewmatrix, p.projmatrix, self.praw, p.campos, self.tanfov[0], self.tanfov[1],
                       L["dL_dcolor"], L["dL_ddepth"], pose_only=True, **self.kw)
        assert g["tau_sum"].data_ptr() == self.packed.data_ptr()
        if self._sharded():  # the abort word joins the sum: any rank's abort reaches every rank
            self.packed[tbs.ABORTED].copy_(self._abort_word)

    def _step(self):
        skip = self.packed[tbs.ABORTED].data_ptr() if self._sharded() else self._abort_ptr
        self.pose.step(self.packed[tbs.TAU], self.packed[tbs.EXPOSURE_GRADS], skip=skip)

    def set_frame(self, gt_color, gt_depth=None, grad_mask=None, w2c=None):
        """New frame: ground truth (device, [3,H,W] / [H,W]; grad_mask [1,H,W] or None) and optionally a new initial pose.
        Everything is copied into buffers the tracker owns, so a captured graph stays valid from frame to frame."""
        if self.gt_color is None:
            self.gt_color = gt_color.to(self.dev, torch.float32).contiguous().clone()
            self.gt_depth = None if gt_depth is None else gt_depth.to(self.dev, torch.float32).contiguous().clone()
            self.grad_mask = None if grad_mask is None else grad_mask.to(self.dev, torch.uint8).contiguous().view(-1).clone()
        else:
            if (gt_depth is None) != (self.gt_depth is None) or (grad_mask is None) != (self.grad_mask is None):
                raise _lib.GsajError("set_frame: depth / mask must be given for every frame of a tracker or for none")
            self.gt_color.copy_(gt_color)
            if gt_depth is not None:
                self.gt_depth.copy_(gt_depth)
            if grad_mask is not None:
                self.grad_mask.copy_(grad_mask.to(self.dev, torch.uint8).view(-1))
        if w2c is not None:
            self.pose.reset(w2c)

    def _sharded(self):
        import torch.distributed as dist
        return self.group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _capture(self):
        ga = torch.cuda.CUDAGraph()
        if self._sharded():  # the collective sits between two graphs
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                self._render_and_grads(False)
            with torch.cuda.graph(gb, pool=ga.pool()):
                self._step()
        else:  # one process: the whole iteration is ONE graph (each graph launch has a cost of its own)
            gb = None
            with torch.cuda.graph(ga):
                self._render_and_grads(False)
                self._step()
        self._graphs = (ga, gb, self.ctx.binning.data_ptr())

    def _eager(self, sync):
        self._render_and_grads(sync)
        tbs.allreduce_pose_terms(self.packed, self.group)
        self._step()

    def iterate(self, n, check_every=0):
        """Up to n tracking iterations; check_every > 0: stop once the device reports |tau| < converged_threshold, looked at
        every check_every iterations (the reference looks every iteration, slam_frontend.py:176-178).  Returns the number run."""
        if self.gt_color is None:
            raise _lib.GsajError("set_frame() first")
        done = 0
        with torch.cuda.device(self.dev):
            while done < n:
                if self.ctx.capacity == 0:
                    self._eager(True)  # the very first iteration sizes the binning arena: synchronous forward
                elif not self.use_graph:
                    self._eager(False)
                else:
                    self.ctx._grow_ahead()  # (a replayed graph calls no forward: the arena watch runs here, outside the capture)
                    if self._graphs is None or self._graphs[2] != self.ctx.binning.data_ptr():
                        self._capture()  # (again if the arena was re-allocated: the graph holds its address)
                    self._graphs[0].replay()
                    if self.ctx.auto_grow:
                        self.ctx.watch.post()
                    if self._graphs[1] is not None:
                        tbs.allreduce_pose_terms(self.packed, self.group)
                        self._graphs[1].replay()
                done += 1
                if check_every and done % check_every == 0 and bool(self.pose.converged.item() != 0.0):
                    break
        self.iterations += done
        err = None
        try:
            self.ctx.status()
        except _lib.GsajError as e:
            err = e
        if self._sharded():  # a rank whose own share fitted must raise with the others (they all skipped the same pose steps)
            import torch.distributed as dist
            flag = torch.tensor([1.0 if err else 0.0], device=self.dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if err is None and float(flag.item()) > 0.0:
                err = _lib.GsajError("an asynchronous forward of another rank's tile band was aborted on the device (binning arena too small)")
        if err is not None:
            # an asynchronous frame did not fit the arena sized earlier (the view moved onto more Gaussians): those iterations
            # rendered nothing and moved nothing (the pose step skips an aborted frame).  The next call re-sizes with one
            # synchronous iteration; the caller sees the error and decides.
            self.ctx.capacity = 0
            raise err
        return done

    @property
    def w2c(self):
        return self.pose.w2c

    @property
    def loss_terms(self):
        """[loss, L_rgb, L_depth] of the last iteration (whole frame: summed over the band shares)."""
        return self.packed[tbs.LOSS_TERMS]
