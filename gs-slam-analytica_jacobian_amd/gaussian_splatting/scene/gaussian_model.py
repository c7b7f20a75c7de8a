"""Parameter store + activations of the Gaussian map: the part of the reference's
GaussianModel that the render path reads (gaussian_splatting/scene/gaussian_model.py:141-177:
get_xyz, get_features, get_opacity, get_scaling, get_rotation, get_covariance,
active_sh_degree / max_sh_degree), plus what sits directly either side of it: the
densification bookkeeping fed by a backward (add_densification_stats :767-771, max_radii2D, n_obs --
one device launch, gsaj_densification_stats) and parameter I/O in the reference's formats
(load_tensors :70-138, save_ply :402-436, load_ply :453-542; gsaj.model_io).  The densify / prune /
optimiser-surgery logic itself (:281-765) stays the caller's.
"""
import torch

from gaussian_splatting.utils.general_utils import build_scaling_rotation, inverse_sigmoid, strip_symmetric


class GaussianModel:
    def __init__(self, sh_degree: int, config=None):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = e
        self._scaling = self._rotation = self._opacity = e
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.covariance_activation = self.build_covariance_from_scaling_rotation
        self.config = config
        self.isotropic = False

    @classmethod
    def from_activated(cls, xyz, scales, rotations, opacities, shs, sh_degree=3, active_sh_degree=None,
                       device="cuda", requires_grad=True):
        """Build a model from *activated* parameters (what the getters will return)."""
        m = cls(sh_degree)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)  # noqa: E731
        m._xyz = t(xyz).clone().requires_grad_(requires_grad)
        shs = t(shs)
        m._features_dc = shs[:, :1, :].clone().contiguous().requires_grad_(requires_grad)
        m._features_rest = shs[:, 1:, :].clone().contiguous().requires_grad_(requires_grad)
        m._scaling = torch.log(t(scales)).requires_grad_(requires_grad)
        m._rotation = t(rotations).clone().requires_grad_(requires_grad)
        m._opacity = inverse_sigmoid(t(opacities)).requires_grad_(requires_grad)
        m.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
        return m

    def build_covariance_from_scaling_rotation(self, scaling, scaling_modifier, rotation):
        L = build_scaling_rotation(scaling_modifier * scaling, rotation)
        return strip_symmetric(L @ L.transpose(1, 2))

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self.rotation_activation(self._rotation))

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    def assign_bucket_gradients(self, g, accumulate=False):
        """.grad of the six raw parameters from the gradients a batched backward leaves in its bucket (gsaj.rasterizer.BatchContext /
        FrameContext: g["mean3D"], g["sh"], g["opacity"], g["scale"], g["rot"] w.r.t. the ACTIVATED quantities the rasteriser is fed):
        what loss.backward() would have left there through the activations of the reference model (exp, normalize, sigmoid, cat;
        gaussian_model.py:41-56, 141-165), in closed form.  accumulate=True adds to existing .grad (several windows per step)."""
        with torch.no_grad():
            P = self._xyz.shape[0]
            sh = g["sh"].view(P, -1, 3)
            op = self.get_opacity
            sc = self.get_scaling
            g_sc = g["scale"] * (sc if sc.shape[1] == 3 else sc.expand(-1, 3))
            if self._scaling.shape[1] == 1:  # isotropic model: one log-scale drives the three axes (gaussian_renderer/__init__.py:98-101)
                g_sc = g_sc.sum(dim=1, keepdim=True)
            q = self._rotation
            n = q.norm(dim=1, keepdim=True).clamp_min(1e-12)
            qh = q / n
            g_q = (g["rot"] - qh * (g["rot"] * qh).sum(dim=1, keepdim=True)) / n
            grads = ((self._xyz, g["mean3D"]), (self._features_dc, sh[:, :1, :]), (self._features_rest, sh[:, 1:, :]),
                     (self._opacity, g["opacity"].view(P, 1) * op * (1.0 - op)), (self._scaling, g_sc), (self._rotation, g_q))
            for prm, gr in grads:
                gr = gr.reshape(prm.shape).to(prm.dtype)
                if accumulate and prm.grad is not None:
                    prm.grad += gr
                else:
                    prm.grad = gr.clone()

    # ---- bookkeeping tensors + parameter I/O -----------------------------------------------------------
    def _init_aux(self):
        """The auxiliary tensors load_tensors / load_ply create (gaussian_model.py:124-131, 538-542)."""
        n, dev = self._xyz.shape[0], self._xyz.device
        self.max_radii2D = torch.zeros((n,), device=dev)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.unique_kfIDs = torch.zeros((n,)).int()
        self.n_obs = torch.zeros((n,)).int()

    def _set_params(self, xyz, f_dc, f_rest, opacity, scaling, rotation, device):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(device).contiguous().requires_grad_(True)  # noqa: E731
        self._xyz, self._features_dc, self._features_rest = t(xyz), t(f_dc), t(f_rest)
        self._opacity, self._scaling, self._rotation = t(opacity), t(scaling), t(rotation)
        self._init_aux()

    def load_tensors(self, model_path, device="cuda"):
        """Reference load_tensors (:70-138): parameters in stored order xyz, f_dc, f_rest, opacity, scaling, rotation; a 2-D f_dc
        becomes [P,1,3]; active_sh_degree is left as it is (the reference never raises it here).  Nothing in the file is
        executed (gsaj.model_io.read_parameter_tensors).  Returns True / False like the reference."""
        from gsaj.model_io import read_parameter_tensors
        try:
            ts = read_parameter_tensors(model_path)
            # the reference takes named_parameters() of the scripted module (parameters only, in registration order); the
            # restricted reader returns every tensor of the pickled state in stored order and does NOT tell parameters from
            # buffers, so the six must be exactly six and must look like xyz, f_dc, f_rest, opacity, scaling, rotation -- an
            # archive with anything else in between would otherwise be mapped to the wrong fields without a word
            if len(ts) != 6:
                raise ValueError("expected exactly 6 parameter tensors (xyz, f_dc, f_rest, opacity, scaling, rotation), found %d" % len(ts))
            xyz, f_dc, f_rest, opacity, scaling, rotation = ts
            if f_dc.dim() == 2:
                f_dc = f_dc.unsqueeze(1)
            P = xyz.shape[0] if xyz.dim() == 2 else -1
            ok = (all(t.is_floating_point() for t in ts) and xyz.dim() == 2 and xyz.shape[1] == 3
                  and f_dc.dim() == 3 and tuple(f_dc.shape[::2]) == (P, 3) and f_dc.shape[1] == 1
                  and f_rest.dim() == 3 and tuple(f_rest.shape[::2]) == (P, 3)
                  and tuple(opacity.shape) == (P, 1)
                  and scaling.dim() == 2 and scaling.shape[0] == P and scaling.shape[1] in (1, 3)
                  and tuple(rotation.shape) == (P, 4))
            if not ok:
                raise ValueError("the 6 tensors do not have the shapes of (xyz [P,3], f_dc [P,1,3], f_rest [P,M-1,3], opacity [P,1], "
                                 "scaling [P,1|3], rotation [P,4]): %s" % ([tuple(t.shape) for t in ts],))
            self._set_params(xyz, f_dc, f_rest, opacity, scaling, rotation, device)
            return True
        except Exception as e:  # noqa: BLE001 -- the reference reports and returns False
            print("Error loading tensors from %s: %s" % (model_path, e))
            return False

    def construct_list_of_attributes(self):
        from gsaj.model_io import ply_attributes
        return ply_attributes(self._features_dc.shape[1] * self._features_dc.shape[2], self._features_rest.shape[1] * self._features_rest.shape[2],
                              self._scaling.shape[1], self._rotation.shape[1])

    def save_ply(self, path):
        import os
        from gsaj.model_io import write_ply
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        d = lambda x: x.detach().cpu().numpy()  # noqa: E731
        write_ply(path, d(self._xyz), d(self._features_dc), d(self._features_rest), d(self._opacity), d(self._scaling), d(self._rotation))

    def load_ply(self, path, device="cuda"):
        from gsaj.model_io import read_gaussian_ply
        g = read_gaussian_ply(path, self.max_sh_degree)
        self._set_params(g["xyz"], g["f_dc"], g["f_rest"], g["opacity"], g["scaling"], g["rotation"], device)
        self.ply_input = dict(points=g["xyz"], normals=g["normals"])
        self.active_sh_degree = self.max_sh_degree

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """Reference :767-771, one device launch: xyz_gradient_accum[f] += ||grad[f, :2]||, denom[f] += 1.  update_filter is the
        view's visibility filter (radii > 0), as every caller passes it (slam_backend.py:119, 282)."""
        self.densification_step(viewspace_point_tensor.grad[None], update_filter[None].to(torch.int32), None, update_max_radii=False)

    def densification_step(self, dL_dmean2D, radii, n_touched=None, update_max_radii=True):
        """All bookkeeping of one mapping iteration over K views in ONE launch: dL_dmean2D [K,P,3] (BatchContext's g["mean2D"] or
        stacked viewspace_points.grad), radii [K,P] int32, n_touched [K,P] int32 or None.  Updates xyz_gradient_accum, denom,
        max_radii2D (visible = radii > 0) and, with n_touched, n_obs = number of views that touched each Gaussian
        (slam_backend.py:113-121, 236-250, 276-285)."""
        from gsaj import _lib
        lib = _lib.load()
        K, P = radii.shape
        g = dL_dmean2D.to(torch.float32).contiguous()
        r = radii.to(torch.int32).contiguous()
        nt = None if n_touched is None else n_touched.to(torch.int32).contiguous()
        dev = g.device
        n_obs = torch.zeros((P,), dtype=torch.int32, device=dev) if nt is not None else None
        if self.max_radii2D.device != dev:
            self.max_radii2D = self.max_radii2D.to(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.gsaj_densification_stats(K, P, g.data_ptr(), r.data_ptr(), None if nt is None else nt.data_ptr(),
                                                    self.xyz_gradient_accum.data_ptr(), self.denom.data_ptr(),
                                                    self.max_radii2D.data_ptr() if update_max_radii else None,
                                                    None if n_obs is None else n_obs.data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream), "gsaj_densification_stats")
        if n_obs is not None:
            self.n_obs = n_obs
        return n_obs
