"""Synthetic cameras and Gaussian scenes (NumPy only, no device code).

Real Replica / TUM / EuRoC data and the reference's optimized_params*.pt blobs are not
available offline, so every benchmark and parity test runs on seeded synthetic scenes of
the shapes SURVEY.md 8(d) fixes.  Camera conventions follow the reference:
utils/camera_utils.py:95-109 (world_view_transform = W2C^T, full_proj_transform =
W2C^T P^T, camera_center = inv(W2C^T)[3,:3]) and
gaussian_splatting/utils/graphics_utils.py:72-93 (getProjectionMatrix2).
"""
import math

import numpy as np

# Jacob_test_result/w2c_gt.txt and T_noise.txt of the reference (data fixtures; the
# rotation block of w2c_gt is 0.4645 x orthonormal -- a similarity transform).
W2C_GT = np.array([
    [7.653097063302993774e-02, 4.581378996372222900e-01, 1.176195684820413589e-03, -4.326294064521789551e-01],
    [2.759018540382385254e-01, -4.513651877641677856e-02, -3.709307014942169189e-01, -2.221532016992568970e-01],
    [-3.657456636428833008e-01, 6.181478872895240784e-02, -2.795670628547668457e-01, 1.228660702705383301e+00],
    [0.0, 0.0, 0.0, 1.0]], dtype=np.float64)
T_NOISE = np.array([
    [9.848077530122081313e-01, -1.736481776669303312e-01, 0.0, 0.15],
    [1.736481776669303312e-01, 9.848077530122081313e-01, 0.0, 0.15],
    [0.0, 0.0, 1.0, 0.15],
    [0.0, 0.0, 0.0, 1.0]], dtype=np.float64)


def projection_matrix(znear, zfar, fx, fy, cx, cy, W, H):
    """P of getProjectionMatrix2 (graphics_utils.py:72-93), closed form."""
    P = np.zeros((4, 4), np.float64)
    P[0, 0] = 2.0 * fx / W
    P[1, 1] = 2.0 * fy / H
    P[0, 2] = (2.0 * cx - W) / W
    P[1, 2] = (2.0 * cy - H) / H
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def orthonormalize(w2c):
    """Nearest rigid transform (drops the similarity scale of W2C_GT)."""
    out = np.array(w2c, np.float64)
    u, _, vt = np.linalg.svd(out[:3, :3])
    R = u @ vt
    if np.linalg.det(R) < 0:
        u[:, -1] *= -1
        R = u @ vt
    out[:3, :3] = R
    return out


def make_camera(w2c, W=640, H=480, fx=577.5, fy=577.5, cx=319.5, cy=239.5, znear=0.01, zfar=100.0):
    """All matrices the rasteriser consumes, as float32 arrays (defaults: the intrinsics
    hard-coded in the reference scripts, Loss_Derivative_script_compare.py:1406-1419)."""
    w2c = np.asarray(w2c, np.float64)
    P = projection_matrix(znear, zfar, fx, fy, cx, cy, W, H)
    f = np.float32
    w2c32 = w2c.astype(f)
    P32 = P.astype(f)
    view = w2c32.T.copy()
    full = (view @ P32.T).astype(f)  # camera_utils.py:99-105, fp32 like torch
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(f)
    return dict(W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, znear=znear, zfar=zfar, w2c=w2c,
                viewmatrix=view, projmatrix=full, projmatrix_raw=P32.T.copy(), campos=campos,
                tanfovx=W / (2.0 * fx), tanfovy=H / (2.0 * fy),
                FoVx=2 * math.atan(W / (2 * fx)), FoVy=2 * math.atan(H / (2 * fy)))


def fixture_camera(noisy=True, orthonormal=False, **kw):
    """Camera of the reference's Jacobian test: w2c = w2c_gt @ T_noise
    (Loss_Derivative_script_compare.py:1424, Jacobian_test.py)."""
    w2c = W2C_GT @ T_NOISE if noisy else W2C_GT.copy()
    if orthonormal:
        w2c = orthonormalize(w2c)
    return make_camera(w2c, **kw)


def random_unit_quaternions(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_scene(P, seed, cam, z_range=(1.0, 6.0), log_scale_range=(math.log(0.005), math.log(0.04)),
               opacity_range=(0.3, 0.95), sh_coeffs=16, margin=0.05, sh_sigma=0.3):
    """P Gaussians placed uniformly in the camera frustum (pixel uniform, depth uniform),
    log-uniform per-axis scales, random rotations, uniform opacities, SH ~ N(0, sh_sigma)
    with DC ~ U(-1, 1).  All parameters are *activated* values (what GaussianModel's
    getters return, gaussian_model.py:151-177)."""
    rng = np.random.default_rng(seed)
    W, H = cam["W"], cam["H"]
    u = rng.uniform(-margin * W, (1 + margin) * W, size=P)
    v = rng.uniform(-margin * H, (1 + margin) * H, size=P)
    z = rng.uniform(z_range[0], z_range[1], size=P)
    xc = (u - cam["cx"]) / cam["fx"] * z
    yc = (v - cam["cy"]) / cam["fy"] * z
    pc = np.stack([xc, yc, z, np.ones(P)], axis=1)
    c2w = np.linalg.inv(cam["w2c"])
    # a similarity W2C (rotation block s*R) maps world lengths to s * camera lengths
    s = np.cbrt(abs(np.linalg.det(cam["w2c"][:3, :3])))
    means = (pc @ c2w.T)[:, :3]
    scales = np.exp(rng.uniform(log_scale_range[0], log_scale_range[1], size=(P, 3))) / s
    rots = random_unit_quaternions(rng, P)
    opac = rng.uniform(opacity_range[0], opacity_range[1], size=(P, 1))
    shs = rng.normal(0.0, sh_sigma, size=(P, sh_coeffs, 3))
    shs[:, 0, :] = rng.uniform(-1.0, 1.0, size=(P, 3))
    f = np.float32
    return dict(means3D=means.astype(f), scales=scales.astype(f), rotations=rots.astype(f),
                opacities=opac.astype(f), shs=shs.astype(f))


def covariance6(scales, rotations, modifier=1.0):
    """Sigma = R S S^T R^T as (xx,xy,xz,yy,yz,zz) (general_utils.py:97-148 build_scaling_rotation
    + strip_symmetric; quaternion (r,x,y,z))."""
    q = rotations.astype(np.float64)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    L = R * (scales.astype(np.float64) * modifier)[:, None, :]
    S = L @ np.transpose(L, (0, 2, 1))
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)


# Named workloads of BASELINE.json `configs` (synthetic stand-ins).
def config_scene(name):
    if name == "cfg1":  # N=15, committed camera
        cam = fixture_camera(noisy=True)
        sc = make_scene(15, 15, cam, z_range=(0.8, 1.6), log_scale_range=(math.log(0.01), math.log(0.05)))
    elif name == "cfg2":  # headline: 50k Gaussians, 640x480
        cam = fixture_camera(noisy=True, orthonormal=True)
        sc = make_scene(50_000, 50000, cam, z_range=(1.0, 6.0))
    elif name == "cfg3":  # Replica calibration, 300k, SH0
        cam = make_camera(orthonormalize(W2C_GT @ T_NOISE), W=1200, H=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5)
        sc = make_scene(300_000, 300000, cam, z_range=(1.0, 5.0), sh_coeffs=1)
    elif name == "cfg4":  # TUM fr1_desk calibration (configs/rgbd/tum/fr1_desk.yaml), 100k Gaussians; the 8 keyframes: config_window
        cam = make_camera(orthonormalize(W2C_GT @ T_NOISE), **TUM_FR1)
        sc = make_scene(100_000, 100000, cam, z_range=(1.0, 6.0), margin=0.25)
    elif name == "cfg5":  # 1M Gaussians, 1280x720
        cam = make_camera(orthonormalize(W2C_GT @ T_NOISE), W=1280, H=720, fx=870.0, fy=870.0, cx=639.5, cy=359.5)
        sc = make_scene(1_000_000, 1000000, cam, z_range=(1.0, 8.0), sh_coeffs=1)
    else:
        raise KeyError(name)
    return cam, sc


TUM_FR1 = dict(W=640, H=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989)


def config_window(name="cfg4", n=8):
    """Mapping window of BASELINE config 4 (synthetic stand-in): n keyframe cameras (window_size 8,
    configs/rgbd/tum/base_config.yaml:36) on a 0.5 m arc over ONE shared Gaussian map -> (cameras, scene)."""
    cam0, sc = config_scene(name)
    kw = {k: cam0[k] for k in ("W", "H", "fx", "fy", "cx", "cy")}
    return keyframe_cameras(n, **kw), sc


def keyframe_cameras(n, radius=0.5, **kw):
    """n poses on an arc looking at the scene (cfg4-synth: TUM fr1 intrinsics 517.3/516.5)."""
    base = orthonormalize(W2C_GT @ T_NOISE)
    cams = []
    for k in range(n):
        a = (k - (n - 1) / 2.0) * (radius / max(n - 1, 1)) * 0.6
        d = np.eye(4)
        d[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        d[0, 3] = -radius * math.sin(a)
        cams.append(make_camera(d @ base, **kw))
    return cams
