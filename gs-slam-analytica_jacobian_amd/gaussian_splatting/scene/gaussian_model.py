"""Parameter store + activations of the Gaussian map: the part of the reference's
GaussianModel that the render path reads (gaussian_splatting/scene/gaussian_model.py:141-177:
get_xyz, get_features, get_opacity, get_scaling, get_rotation, get_covariance,
active_sh_degree / max_sh_degree).  Densification, pruning, optimiser surgery, PLY and
TorchScript I/O (:70-138, :281-771) are callers' business and out of scope here.
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
