"""CPU: parameter I/O of the Gaussian map in the reference's formats -- gsaj.model_io + the overlay GaussianModel.

PLY: layout and property order of the reference's save_ply / construct_list_of_attributes (gaussian_model.py:383-436), read back
by the restatement of load_ply (:453-542).  The reference writes through `plyfile`, which is not installed here (and cannot be):
byte-level identity with plyfile's output is parity unpinned; the header grammar follows the PLY specification and the
reference's attribute list, and ascii / big-endian bodies are read too.
.pt: load_tensors (:70-138) takes the parameters of a TorchScript module in registration order; here such an archive is
written with torch.jit (module defined below) and read WITHOUT torch.jit.load (nothing in the file is executed)."""
import os
import pickle

import numpy as np
import pytest
import torch

from gaussian_splatting.scene.gaussian_model import GaussianModel
from gsaj import model_io as mio


class SixParams(torch.nn.Module):
    """Stand-in for the module behind optimized_params*.pt: six parameters in the reference's order, f_dc stored 2-D."""

    def __init__(self, tensors):
        super().__init__()
        for name, t in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), tensors):
            self.register_parameter(name, torch.nn.Parameter(t))

    def forward(self):
        return self.xyz


def _model(P=41, seed=0):
    rng = np.random.default_rng(seed)
    return GaussianModel.from_activated(rng.normal(size=(P, 3)), rng.uniform(0.01, 0.1, (P, 3)), rng.normal(size=(P, 4)),
                                        rng.uniform(0.1, 0.9, (P, 1)), rng.normal(size=(P, 16, 3)), device="cpu")


def test_ply_round_trip_and_layout(tmp_path):
    m = _model()
    path = str(tmp_path / "sub" / "map.ply")
    m.save_ply(path)
    head = open(path, "rb").read(2000).split(b"end_header")[0].decode()
    names = [ln.split()[2] for ln in head.splitlines() if ln.startswith("property")]
    assert names == m.construct_list_of_attributes() == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
                                                         + ["f_rest_%d" % i for i in range(45)] + ["opacity"]
                                                         + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)])
    assert "format binary_little_endian 1.0" in head and "element vertex 41" in head
    m2 = GaussianModel(3)
    m2.load_ply(path, device="cpu")
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a.detach(), b.detach()) and b.requires_grad
    assert m2.active_sh_degree == 3 and m2.max_radii2D.shape == (41,) and m2.n_obs.dtype == torch.int32
    # channel-major feature storage, as save_ply's transpose(1, 2).flatten(start_dim=1) produces
    v = mio.read_ply_vertices(path)
    assert np.allclose(v["f_rest_0"], m._features_rest.detach().numpy()[:, 0, 0]) and np.allclose(v["f_rest_15"], m._features_rest.detach().numpy()[:, 0, 1])
    # ascii variant of the same data
    cols = np.stack([v[n] for n in names], axis=1)
    apath = str(tmp_path / "ascii.ply")
    with open(apath, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex 41\n" + "".join("property float %s\n" % n for n in names) + "end_header\n")
        np.savetxt(fh, cols, fmt="%.9g")
    m3 = GaussianModel(3)
    m3.load_ply(apath, device="cpu")
    assert torch.allclose(m3._xyz.detach(), m._xyz.detach(), rtol=1e-6) and torch.allclose(m3._rotation.detach(), m._rotation.detach(), rtol=1e-6)
    with pytest.raises(ValueError):
        GaussianModel(2).load_ply(path, device="cpu")  # SH degree mismatch: the reference asserts too


def test_load_tensors_reads_torchscript_archives_without_running_them(tmp_path):
    m = _model(seed=3)
    ps = [p.detach().clone() for p in m.parameters()]
    ps[1] = ps[1].squeeze(1)  # 2-D f_dc: load_tensors restores [P,1,3] (gaussian_model.py:103-106)
    path = str(tmp_path / "optimized_params_small.pt")
    torch.jit.script(SixParams(ps)).save(path)
    assert [tuple(t.shape) for t in mio.read_parameter_tensors(path)] == [(41, 3), (41, 3), (41, 15, 3), (41, 1), (41, 3), (41, 4)]
    m2 = GaussianModel(3)
    assert m2.load_tensors(path, device="cpu") is True
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a.detach(), b.detach())
    assert m2.active_sh_degree == 0 and m2.xyz_gradient_accum.shape == (41, 1) and m2.denom.shape == (41, 1)
    # plain torch.save of a list goes through the weights-only loader
    p2 = str(tmp_path / "list.pt")
    torch.save([p.detach() for p in m.parameters()], p2)
    m3 = GaussianModel(3)
    assert m3.load_tensors(p2, device="cpu") and torch.equal(m3._scaling.detach(), m._scaling.detach())
    # missing file / too few tensors: reported, False (the reference's behaviour)
    assert GaussianModel(3).load_tensors(str(tmp_path / "nope.pt"), device="cpu") is False
    torch.save([ps[0]], str(tmp_path / "short.pt"))
    assert GaussianModel(3).load_tensors(str(tmp_path / "short.pt"), device="cpu") is False
    # an archive whose tensors are not exactly the six parameters (a buffer in between, or a seventh tensor) would be mapped to the
    # wrong fields position by position: it is refused instead (the restricted reader cannot tell buffers from parameters)
    full = [p.detach() for p in m.parameters()]
    torch.save(full + [torch.zeros(3)], str(tmp_path / "seven.pt"))
    assert GaussianModel(3).load_tensors(str(tmp_path / "seven.pt"), device="cpu") is False
    torch.save([full[0], torch.zeros(41, 7), *full[1:5]], str(tmp_path / "shifted.pt"))  # six tensors, one of them a stranger
    assert GaussianModel(3).load_tensors(str(tmp_path / "shifted.pt"), device="cpu") is False


def test_parameter_reader_refuses_foreign_callables(tmp_path):
    """A data.pkl that names anything but tensor-rebuild helpers / containers / module classes is rejected, not executed."""
    import io
    import zipfile

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    path = str(tmp_path / "evil.pt")
    with zipfile.ZipFile(path, "w") as zf:
        zf.writestr("archive/data.pkl", pickle.dumps(Evil()))
        zf.writestr("archive/constants.pkl", pickle.dumps(()))
        zf.writestr("archive/code/__torch__.py", "")
    with pytest.raises(pickle.UnpicklingError):
        mio.read_parameter_tensors(path)
    assert GaussianModel(3).load_tensors(path, device="cpu") is False
    assert not (tmp_path / "pwned").exists()
