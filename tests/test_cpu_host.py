"""CPU: the C-ABI library loads and exports every declared symbol; host-side mirrors of the
reference interface (pose_utils, camera matrices, SH utils, GaussianModel getters, argument
errors); multi-process keyframe sharding over gloo (world_size 2).  No kernel is launched."""
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gsaj import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "gsaj.h")) as fh:
        declared = set(re.findall(r"\b(gsaj_[a-z0-9_]+)\s*\(", fh.read()))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gsaj_version() >= 100
    # pure host queries work without a GPU and are monotone
    assert lib.gsaj_geom_workspace_bytes(1000) < lib.gsaj_geom_workspace_bytes(2000)
    assert lib.gsaj_binning_workspace_bytes(1000) < lib.gsaj_binning_workspace_bytes(100000)
    assert lib.gsaj_image_workspace_bytes(640, 480) >= 640 * 480 * 8
    assert lib.gsaj_dense_workspace_bytes(15, 640, 480) >= 1200 * 15 * 48
    # argument errors are reported through return code + gsaj_last_error (no GPU needed)
    assert lib.gsaj_forward_preprocess(0, 0, 0, 640, 480, *([None] * 5), 1.0, *([None] * 5), 1.0, 1.0, 0, *([None] * 5)) == -1
    assert b"invalid argument" in lib.gsaj_last_error()


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from gsaj import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gs-slam-analytica_jacobian_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as fh:
                    txt = fh.read()
                assert "oracle" not in txt.replace("oracle sorts", "").replace("CPU oracle", ""), os.path.join(dirpath, fn)


def test_cpu_tensors_are_rejected():
    """There is no CPU product path: the op refuses host tensors instead of falling back."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    s = GaussianRasterizationSettings(48, 64, 0.5, 0.4, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4), 0,
                                      torch.zeros(3), False, False)
    r = GaussianRasterizer(s)
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(m, m.clone(), torch.ones(4, 1), colors_precomp=torch.ones(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m.clone(), torch.ones(4, 1), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m.clone(), torch.ones(4, 1), colors_precomp=torch.ones(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4),
          cov3D_precomp=torch.ones(4, 6))


def test_pose_utils_against_matrix_exponential():
    from utils.pose_utils import SE3_exp, SO3_exp, V, skew_sym_mat, rt2mat, update_pose

    g = torch.Generator().manual_seed(0)
    for scale in (1e-7, 1e-3, 0.3, 2.0):
        tau = (torch.randn(6, generator=g, dtype=torch.float64) * scale)
        xi = torch.zeros(4, 4, dtype=torch.float64)
        xi[:3, :3] = skew_sym_mat(tau[3:])
        xi[:3, 3] = tau[:3]
        assert torch.allclose(SE3_exp(tau), torch.matrix_exp(xi), atol=1e-12)
        assert torch.allclose(SO3_exp(tau[3:]) @ SO3_exp(tau[3:]).T, torch.eye(3, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(V(torch.zeros(3)), torch.eye(3))
    assert rt2mat(np.eye(3), np.array([1, 2, 3.0]))[:3, 3].tolist() == [1, 2, 3]

    class Cam:
        def __init__(self):
            self.R, self.T = torch.eye(3), torch.zeros(3)
            self.cam_rot_delta = torch.nn.Parameter(torch.tensor([0.0, 0.1, 0.0]))
            self.cam_trans_delta = torch.nn.Parameter(torch.tensor([0.2, 0.0, 0.0]))

        def update_RT(self, R, t):
            self.R, self.T = R, t

    c = Cam()
    assert not bool(update_pose(c))
    assert float(c.cam_rot_delta.abs().sum()) == 0 and float(c.cam_trans_delta.abs().sum()) == 0
    assert torch.allclose(c.R, SO3_exp(torch.tensor([0.0, 0.1, 0.0])), atol=1e-6)
    assert bool(update_pose(c))  # zero delta -> converged


def test_camera_matrices_match_synthetic_and_reference_conventions():
    from gsaj import synthetic as syn
    from utils.camera_utils import Camera
    from gaussian_splatting.utils.graphics_utils import focal2fov, fov2focal, getProjectionMatrix2

    cam = syn.fixture_camera(noisy=True)
    v = Camera.from_synthetic(cam, device="cpu")
    assert torch.allclose(v.world_view_transform, torch.tensor(cam["viewmatrix"]), atol=1e-6)
    assert torch.allclose(v.full_proj_transform, torch.tensor(cam["projmatrix"]), atol=1e-5)
    assert torch.allclose(v.camera_center, torch.tensor(cam["campos"]), atol=1e-4)
    P = getProjectionMatrix2(0.01, 100.0, 319.5, 239.5, 577.5, 577.5, 640, 480)
    assert torch.allclose(P.T, torch.tensor(cam["projmatrix_raw"]), atol=1e-7)
    assert abs(P[0, 0] - 2 * 577.5 / 640) < 1e-7 and abs(P[0, 2] - (2 * 319.5 - 640) / 640) < 1e-7 and P[3, 2] == 1
    assert abs(fov2focal(focal2fov(577.5, 640), 640) - 577.5) < 1e-9
    assert abs(math.tan(v.FoVx * 0.5) - cam["tanfovx"]) < 1e-12


def test_sh_utils_and_model_getters():
    from gaussian_splatting.scene.gaussian_model import GaussianModel
    from gaussian_splatting.utils.general_utils import build_rotation, build_scaling_rotation, strip_symmetric
    from gaussian_splatting.utils.sh_utils import RGB2SH, SH2RGB, eval_sh
    from gsaj import synthetic as syn
    from oracle import dense_oracle as dor

    rng = np.random.default_rng(0)
    sh = rng.normal(size=(7, 16, 3))
    d = rng.normal(size=(7, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for deg in range(4):
        want = np.einsum("nk,nkc->nc", dor.sh_basis(deg, d)[:, :16], sh)
        got = eval_sh(deg, torch.tensor(sh).transpose(1, 2), torch.tensor(d)).numpy()
        assert np.allclose(got, want, atol=1e-12)
    assert torch.allclose(SH2RGB(RGB2SH(torch.tensor([0.2, 0.7]))), torch.tensor([0.2, 0.7]))
    cam = syn.fixture_camera()
    sc = syn.make_scene(11, 0, cam)
    m = GaussianModel.from_activated(sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["shs"], device="cpu")
    assert np.allclose(m.get_scaling.detach().numpy(), sc["scales"], rtol=1e-5)
    assert np.allclose(m.get_opacity.detach().numpy(), sc["opacities"], atol=1e-6)
    assert m.get_features.shape == (11, 16, 3) and m.get_rotation.shape == (11, 4)
    assert np.allclose(m.get_covariance().detach().numpy(), syn.covariance6(sc["scales"], sc["rotations"]), rtol=2e-4, atol=1e-9)
    R = build_rotation(torch.tensor(sc["rotations"]))
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(11, 3, 3), atol=1e-5)
    L = build_scaling_rotation(torch.tensor(sc["scales"]), torch.tensor(sc["rotations"]))
    assert strip_symmetric(L @ L.transpose(1, 2)).shape == (11, 6)


def test_loss_functions_seed_values():
    from utils.slam_utils import compute_loss, get_loss_mapping, get_loss_tracking

    class V:
        pass

    v = V()
    H, W = 6, 8
    g = torch.Generator().manual_seed(1)
    v.original_image = torch.rand(3, H, W, generator=g)
    v.depth = torch.rand(H, W, generator=g).numpy() + 0.5
    v.grad_mask = torch.ones(1, H, W, dtype=torch.bool)
    v.exposure_a, v.exposure_b = torch.zeros(1), torch.zeros(1)
    cfg = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
    img = torch.rand(3, H, W, generator=g, requires_grad=True)
    dep = torch.rand(1, H, W, generator=g) + 0.5
    op = torch.ones(1, H, W)
    lt = get_loss_tracking(cfg, img, dep, op, v)
    want = 0.9 * (img - v.original_image).abs().mean() + 0.1 * (dep - torch.tensor(v.depth)[None]).abs().mean()
    assert torch.allclose(lt, want, atol=1e-6)
    lm = get_loss_mapping(cfg, img, dep, v, op)
    assert torch.allclose(lm, want, atol=1e-6)
    lt.backward()
    assert torch.allclose(img.grad, 0.9 * torch.sign(img - v.original_image).detach() / img.numel(), atol=1e-7)

    class M:
        get_scaling = torch.tensor([[1.0, 2.0, 3.0]])

    mask = torch.ones(H, W, dtype=torch.bool)
    l = compute_loss(M, img.detach(), dep, v.original_image, torch.tensor(v.depth), mask)
    want = (img.detach() - v.original_image).abs().mean() + (dep[0] - torch.tensor(v.depth)).abs().mean() + 10 * (2 / 3)
    assert torch.allclose(l, want, atol=1e-5)


def test_synthetic_scene_stats():
    from gsaj import synthetic as syn
    from oracle import oracle as orc

    cam, sc = syn.config_scene("cfg1")
    assert sc["means3D"].shape == (15, 3) and sc["shs"].shape == (15, 16, 3)
    assert np.allclose(np.linalg.norm(sc["rotations"], axis=1), 1, atol=1e-6)
    out, st = orc.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"],
                          cam["tanfovy"], 640, 480, np.zeros(3), shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"],
                          sh_degree=3)
    assert (out["radii"] > 0).all()
    assert np.allclose(out["opacity"], 1 - st["final_T"][None])
    R = syn.orthonormalize(syn.W2C_GT)[:3, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    assert len(syn.keyframe_cameras(8)) == 8


def test_keyframe_shard_bucket_layout():
    from gsaj import keyframe_shard as ks

    P, M = 10, 16
    b = torch.arange(ks.bucket_numel(P, M), dtype=torch.float32)
    v = ks.bucket_views(b, P, M)
    assert v["mean3D"].shape == (P, 3) and v["sh"].shape == (P, 48) and v["rot"].shape == (P, 4)
    assert ks.bucket_numel(P, M) == P * 59
    v2 = ks.bucket_views(torch.zeros(ks.bucket_numel(P, M, True, 8)), P, M, True, 8)
    assert v2["tau_all"].shape == (8, 6)
    v["rot"].zero_()
    assert float(b[-P * 4:].abs().sum()) == 0  # views alias the bucket
    assert ks.shard_keyframes(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((ks.shard_keyframes(10, 4, r) for r in range(4)), [])) == list(range(10))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gsaj import keyframe_shard as ks
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank, K, P, M = dist.get_rank(), 5, 7, 4
mine = ks.shard_keyframes(K, 2, rank)
bucket = torch.zeros(ks.bucket_numel(P, M))
v = ks.bucket_views(bucket, P, M)
tau = torch.zeros(len(mine), 6)
for i, k in enumerate(mine):            # stand-in for forward+backward of keyframe k
    v["mean3D"] += (k + 1)
    v["sh"] += 10 * (k + 1)
    tau[i] = torch.arange(6) + 100 * k
ks.allreduce_gaussian_grads(bucket)
allt = ks.gather_pose_grads(tau, K)
tot = sum(k + 1 for k in range(K))
assert torch.all(v["mean3D"] == tot) and torch.all(v["sh"] == 10 * tot) and torch.all(v["rot"] == 0)
for k in range(K):
    assert torch.equal(allt[k], torch.arange(6) + 100.0 * k), (k, allt[k])
# pose gradients riding in the bucket tail: the sum all-reduce doubles as the all-gather
b2 = torch.zeros(ks.bucket_numel(P, M, True, K))
v2 = ks.bucket_views(b2, P, M, True, K)
for k in mine:
    v2["tau_all"][k] = torch.arange(6) + 100.0 * k
work = ks.allreduce_gaussian_grads(b2, async_op=True)
work.wait()
for k in range(K):
    assert torch.equal(v2["tau_all"][k], torch.arange(6) + 100.0 * k)
# batched windows (gsaj.rasterizer.BatchContext(n_windows=world, window=rank)): K rows per rank in the tail, the others zero
KW = 3
b3 = torch.zeros(ks.bucket_numel(P, M, True, KW * 2))
v3 = ks.bucket_views(b3, P, M, True, KW * 2)
v3["mean3D"][:] = float(rank + 1)
v3["tau_all"][rank * KW:(rank + 1) * KW] = torch.arange(KW * 6, dtype=torch.float32).view(KW, 6) + 1000.0 * rank
ks.allreduce_gaussian_grads(b3)
assert torch.equal(v3["mean3D"], torch.full_like(v3["mean3D"], 3.0))
for r in range(2):
    assert torch.equal(v3["tau_all"][r * KW:(r + 1) * KW], torch.arange(KW * 6, dtype=torch.float32).view(KW, 6) + 1000.0 * r)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_keyframe_shard_two_processes_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    pkg = os.path.join(ROOT, "gs-slam-analytica_jacobian_amd")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
