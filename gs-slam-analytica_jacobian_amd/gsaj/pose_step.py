"""Tracking pose step on the device (C ABI gsaj_pose_adam_step).

`PoseTracker` holds what slam_frontend.tracking keeps in a Camera + torch.optim.Adam (slam_frontend.py:135-193,
utils/pose_utils.py:76-93, camera_utils.py:95-109): the pose W2C, the Adam moments of (cam_trans_delta, cam_rot_delta,
exposure_a, exposure_b) and the exposure values -- as one small device buffer.  `step()` is Adam.step() + update_pose()
in a single launch; the matrices the next render needs are views into the same buffer, and `converged` is a device
scalar, so N tracking iterations can be enqueued without a host round trip."""
import torch

from . import _lib

_O = dict(w2c=(0, 16), m=(16, 24), v=(24, 32), step=(32, 33), exposure=(33, 35), view=(35, 51), proj=(51, 67), campos=(67, 70),
          tau=(70, 76), norm=(76, 77), conv=(77, 78))


class PoseTracker:
    def __init__(self, w2c, projection_matrix, device, lr_rot=0.003, lr_trans=0.001, lr_exposure_a=0.01, lr_exposure_b=0.01,
                 betas=(0.9, 0.999), eps=1e-8, converged_threshold=1e-4):
        self.lib = _lib.load()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GsajError("PoseTracker needs a HIP device (there is no CPU path)")
        n = self.lib.gsaj_pose_state_floats()
        self.state = torch.zeros(n, dtype=torch.float32, device=self.dev)
        self.projection = torch.as_tensor(projection_matrix, dtype=torch.float32).reshape(4, 4).contiguous().to(self.dev)
        self.lr = (float(lr_rot), float(lr_trans), float(lr_exposure_a), float(lr_exposure_b))
        self.betas, self.eps, self.thr = (float(betas[0]), float(betas[1])), float(eps), float(converged_threshold)
        self.reset(w2c)

    def reset(self, w2c, exposure=(0.0, 0.0)):
        """A new frame: pose, zero Adam moments and step count -- IN PLACE, so that views of the state handed out earlier
        (and hipGraphs captured over them, gsaj.tracking) stay valid."""
        self.state.zero_()
        self.state[0:16] = torch.as_tensor(w2c, dtype=torch.float32).reshape(16).to(self.dev)
        self.state[33] = float(exposure[0])
        self.state[34] = float(exposure[1])
        # matrices for the first render
        w = self.state[0:16].view(4, 4)
        self.state[35:51] = w.t().reshape(16)
        self.state[51:67] = (w.t() @ self.projection).reshape(16)
        self.state[67:70] = -(torch.linalg.inv(w[:3, :3].double().cpu()) @ w[:3, 3].double().cpu()).float().to(self.dev)

    def _v(self, name, shape=None):
        a, b = _O[name]
        t = self.state[a:b]
        return t.view(*shape) if shape else t

    w2c = property(lambda s: s._v("w2c", (4, 4)))
    viewmatrix = property(lambda s: s._v("view", (4, 4)))        # world_view_transform = W2C^T
    projmatrix = property(lambda s: s._v("proj", (4, 4)))        # full_proj_transform
    campos = property(lambda s: s._v("campos"))
    exposure_a = property(lambda s: s.state[33:34])
    exposure_b = property(lambda s: s.state[34:35])
    tau = property(lambda s: s._v("tau"))
    converged = property(lambda s: s.state[77])                   # device scalar: 1.0 when |tau| < threshold

    def step(self, dL_dtau_sum, dL_dexposure=None, skip=None):
        """dL_dtau_sum: device float32 [6] = [rho, theta] (gsaj_rasterize_backward); dL_dexposure: device [2] or None.
        skip: device address (int) of a 32-bit word, or None: non-zero when the gradients are stale because their frame was aborted
        on the device (FrameContext.abort_flag_ptr) -- the step then changes nothing."""
        for t in (dL_dtau_sum, dL_dexposure):
            if t is not None and (t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous()):
                raise _lib.GsajError("gradients must be contiguous float32 device tensors")
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(self.lib.gsaj_pose_adam_step(dL_dtau_sum.data_ptr(), None if dL_dexposure is None else dL_dexposure.data_ptr(),
                                                self.lr[0], self.lr[1], self.lr[2], self.lr[3], self.betas[0], self.betas[1], self.eps,
                                                self.thr, self.projection.data_ptr(), self.state.data_ptr(), skip, st),
                   "gsaj_pose_adam_step")


class PoseTrackerBatch:
    """K poses stepped in one launch (C ABI gsaj_pose_adam_step_batch): the keyframe poses of a mapping window, each with its own
    Adam state (utils/slam_backend.py:255-262 steps the keyframe optimiser and calls update_pose per keyframe).  `state` is
    [K, 80]; the matrices the next batched render needs come out as [K,4,4] / [K,3] views of it gathered by `matrices()`."""

    def __init__(self, w2cs, projection_matrix, device, lr_rot=0.003, lr_trans=0.001, lr_exposure_a=0.01, lr_exposure_b=0.01,
                 betas=(0.9, 0.999), eps=1e-8, converged_threshold=1e-4):
        self.lib = _lib.load()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GsajError("PoseTrackerBatch needs a HIP device (there is no CPU path)")
        self.singles = [PoseTracker(w, projection_matrix, device, lr_rot, lr_trans, lr_exposure_a, lr_exposure_b, betas, eps,
                                    converged_threshold) for w in w2cs]  # (initialises every state exactly as the single tracker does)
        self.K = len(self.singles)
        self.state = torch.stack([s.state for s in self.singles]).contiguous()
        self.projection = self.singles[0].projection
        self.lr, self.betas, self.eps, self.thr = self.singles[0].lr, self.singles[0].betas, self.singles[0].eps, self.singles[0].thr
        del self.singles

    def step(self, dL_dtau, dL_dexposure=None, active=None, skip=None, skip_stride=0):
        """dL_dtau [K,6] (BatchContext's g["tau_all"]), dL_dexposure [K,2] or None, active: [K] bool / uint8 or None.
        skip / skip_stride: BatchContext.abort_flags() -- views aborted on the device keep their pose and Adam state."""
        for t, shape in ((dL_dtau, (self.K, 6)), (dL_dexposure, (self.K, 2))):
            if t is not None and (t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape):
                raise _lib.GsajError("gradients must be contiguous float32 device tensors of shape [K,6] / [K,2]")
        act = None if active is None else active.to(device=self.dev, dtype=torch.uint8).contiguous()
        _lib.check(self.lib.gsaj_pose_adam_step_batch(self.K, dL_dtau.data_ptr(), None if dL_dexposure is None else dL_dexposure.data_ptr(),
                                                      None if act is None else act.data_ptr(), self.lr[0], self.lr[1], self.lr[2],
                                                      self.lr[3], self.betas[0], self.betas[1], self.eps, self.thr,
                                                      self.projection.data_ptr(), self.state.data_ptr(), skip, int(skip_stride),
                                                      torch.cuda.current_stream(self.dev).cuda_stream), "gsaj_pose_adam_step_batch")

    def matrices(self):
        """(viewmatrices [K,4,4], projmatrices [K,4,4], campos [K,3]) as contiguous tensors for BatchContext.forward / backward."""
        s = self.state
        return s[:, 35:51].reshape(self.K, 4, 4).contiguous(), s[:, 51:67].reshape(self.K, 4, 4).contiguous(), s[:, 67:70].contiguous()

    w2c = property(lambda s: s.state[:, 0:16].view(s.K, 4, 4))
    exposure = property(lambda s: s.state[:, 33:35])
    converged = property(lambda s: s.state[:, 77])
