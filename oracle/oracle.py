"""ctypes front-end of the CPU oracle (oracle/gsaj_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (gs-slam-analytica_jacobian_amd/) never does.

The two entry points mirror the stages of the reference's rasteriser
(submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-393 forward,
:395-515 backward) on NumPy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libgsaj_oracle.so")
_lib = None

_f = np.float32


def build(force=False):
    """Compile oracle/gsaj_oracle.c with gcc (building the checker is not using it)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, f)) for f in ("gsaj_oracle.c", "chain_body.inc")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.gsaj_oracle_preprocess.restype = ctypes.c_int
        _lib.gsaj_oracle_bin.restype = ctypes.c_int
        _lib.gsaj_oracle_render.restype = ctypes.c_long
        _lib.gsaj_oracle_render_backward.restype = None
        _lib.gsaj_oracle_preprocess_backward.restype = None
        _lib.gsaj_oracle_preprocess_backward_f64.restype = None
        _lib.gsaj_oracle_mark_visible.restype = None
        _lib.gsaj_oracle_set_threads.restype = ctypes.c_int
        _lib.gsaj_oracle_error_model.restype = None
    return _lib


def set_threads(n):
    """OpenMP threads of the oracle's per-Gaussian / per-tile loops (0 = every host core); results do not depend on
    it.  Returns the count in use."""
    return int(_load().gsaj_oracle_set_threads(ctypes.c_int(int(n))))


def round_to_half(a):
    """fp32 -> fp16 (round to nearest even) -> fp32: the rounding the fp16-storage instance records apply to conic,
    opacity and colour (GSAJ_FWD_RECORDS_FP16; __floats2half2_rn on the device)."""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _cf(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=_f)
    if shape is not None:
        a = a.reshape(shape)
    return a


def mark_visible(means3D, viewmatrix):
    lib = _load()
    means3D = _cf(means3D)
    vm = _cf(viewmatrix).reshape(16)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib.gsaj_oracle_mark_visible(ctypes.c_int(P), _p(means3D), _p(vm), _p(out))
    return out.astype(bool)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            sh_degree=0, scale_modifier=1.0, prefiltered=False, record_bits=32):
    """Tiled forward.  viewmatrix / projmatrix are the 4x4 *transposed* matrices the
    reference passes (W2C^T and (P W2C)^T, gaussian_renderer/__init__.py:59-67); flattened
    row-major they are the column-major W2C / P*W2C the kernels index.

    Returns (outputs, state): outputs = colour [3,H,W], depth [1,H,W], opacity [1,H,W],
    radii [P] i32, n_touched [P] i32, num_rendered; state holds every intermediate the
    backward (and the GPU parity tests) need.

    record_bits=16 restates the product's fp16-storage record mode (BASELINE config 5, "fp16 splat with fp32 Jacobian
    accumulation"; the reference has no such mode): the compositor and its backward see conic, opacity and colour rounded
    to half once, everything else (positions, depth, binning, the per-Gaussian chain) is unchanged fp32."""
    lib = _load()
    means3D = _cf(means3D)
    P = means3D.shape[0]
    opac = _cf(opacities).reshape(P)
    vm = _cf(viewmatrix).reshape(16)
    pm = _cf(projmatrix).reshape(16)
    cp = _cf(campos).reshape(3)
    bg = _cf(bg).reshape(3)
    shs = _cf(shs)
    M = 0 if shs is None else shs.shape[1]
    cols = _cf(colors_precomp)
    scales = _cf(scales)
    rots = _cf(rotations)
    covp = _cf(cov3D_precomp)
    if (shs is None) == (cols is None):
        raise ValueError("provide exactly one of shs / colors_precomp")
    if ((scales is None or rots is None) and covp is None) or ((scales is not None or rots is not None) and covp is not None):
        raise ValueError("provide exactly one of (scales, rotations) / cov3D_precomp")

    st = dict(P=P, D=sh_degree, M=M, W=W, H=H)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), _f)
    st["depths"] = np.zeros(P, _f)
    st["cov3D"] = np.zeros((P, 6), _f)
    st["conic_opacity"] = np.zeros((P, 4), _f)
    st["rgb"] = np.zeros((P, 3), _f)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["tiles_touched"] = np.zeros(P, np.int32)
    R = lib.gsaj_oracle_preprocess(
        ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(M), ctypes.c_int(W), ctypes.c_int(H),
        _p(means3D), _p(shs), _p(cols), _p(opac), _p(scales), ctypes.c_float(scale_modifier), _p(rots), _p(covp),
        _p(vm), _p(pm), _p(cp), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), ctypes.c_int(int(prefiltered)),
        _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]), _p(st["conic_opacity"]),
        _p(st["rgb"]), _p(st["clamped"]), _p(st["tiles_touched"]))
    if R < 0:
        raise RuntimeError("Point is filtered although prefiltered is set")
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st["num_rendered"] = R
    st["point_list"] = np.zeros(max(R, 1), np.uint32)
    st["keys"] = np.zeros(max(R, 1), np.uint64)
    st["ranges"] = np.zeros((gx * gy, 2), np.int32)
    rc = lib.gsaj_oracle_bin(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(R), _p(st["radii"]),
                             _p(st["means2D"]), _p(st["depths"]), _p(st["point_list"]), _p(st["keys"]),
                             _p(st["ranges"]))
    if rc != 0:
        raise RuntimeError("oracle binning failed: %d" % rc)
    st["point_list"] = st["point_list"][:R]
    st["keys"] = st["keys"][:R]
    color = np.zeros((3, H, W), _f)
    depth = np.zeros((1, H, W), _f)
    opacity = np.zeros((1, H, W), _f)
    st["final_T"] = np.zeros((H, W), _f)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    n_touched = np.zeros(P, np.int32)
    feats = cols if cols is not None else st["rgb"]
    st["record_bits"] = record_bits
    if record_bits == 16:
        st["conic_opacity_fp32"] = st["conic_opacity"]
        st["conic_opacity"] = round_to_half(st["conic_opacity"])
        feats = round_to_half(feats)
    st["features"] = feats
    pl = st["point_list"] if R > 0 else np.zeros(1, np.uint32)
    st["interactions"] = int(lib.gsaj_oracle_render(
        ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"]), _p(pl), _p(st["means2D"]), _p(feats),
        _p(st["conic_opacity"]), _p(st["depths"]), _p(bg), _p(color), _p(depth), _p(opacity), _p(st["final_T"]),
        _p(st["n_contrib"]), _p(n_touched)))
    st["inputs"] = dict(means3D=means3D, shs=shs, colors_precomp=cols, scales=scales, rotations=rots,
                        cov3D_precomp=covp, viewmatrix=vm, projmatrix=pm, campos=cp, bg=bg,
                        tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier)
    out = dict(color=color, depth=depth, opacity=opacity, radii=st["radii"].copy(), n_touched=n_touched,
               num_rendered=R)
    return out, st


def backward(st, dL_dcolor_img, dL_ddepth_img, projmatrix_raw):
    """Tiled backward from per-pixel seeds dL/dC [3,H,W] and dL/dD [1,H,W].
    projmatrix_raw is P^T (gaussian_renderer/__init__.py:67)."""
    lib = _load()
    P, D, M, W, H = st["P"], st["D"], st["M"], st["W"], st["H"]
    inp = st["inputs"]
    dLc = _cf(dL_dcolor_img).reshape(3, H, W)
    dLd = _cf(dL_ddepth_img).reshape(H, W)
    praw = _cf(projmatrix_raw).reshape(16)
    g = dict(
        dL_dmean2D=np.zeros((P, 3), _f), dL_dconic=np.zeros((P, 2, 2), _f), dL_dopacity=np.zeros((P, 1), _f),
        dL_dcolor=np.zeros((P, 3), _f), dL_ddepth=np.zeros((P, 1), _f))
    feats = st["features"]
    pl = st["point_list"] if st["num_rendered"] > 0 else np.zeros(1, np.uint32)
    lib.gsaj_oracle_render_backward(
        ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"]), _p(pl), _p(st["means2D"]),
        _p(st["conic_opacity"]), _p(feats), _p(st["depths"]), _p(inp["bg"]), _p(st["final_T"]), _p(st["n_contrib"]),
        _p(dLc), _p(dLd), _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolor"]),
        _p(g["dL_ddepth"]))
    g.update(chain(st, g["dL_dmean2D"], g["dL_dconic"], g["dL_dcolor"], g["dL_ddepth"], projmatrix_raw))
    return g


def error_model(st, dL_dcolor_img, dL_ddepth_img, border_rel=1e-5, border_rel_T=1e-4):
    """How far two correct fp32 evaluations of the compositor may differ on this frame (gsaj_oracle_error_model):
    -> dict(term_mass [P,10], cond_slack [P,10], flip_budget [P,10], border_mask [H,W] bool).  Component order:
    mean2D x,y, conic a,b,c, opacity, colour r,g,b, depth."""
    lib = _load()
    P, W, H = st["P"], st["W"], st["H"]
    dLc = _cf(dL_dcolor_img).reshape(3, H, W)
    dLd = _cf(dL_ddepth_img).reshape(H, W)
    out = dict(term_mass=np.zeros((P, 10), _f), cond_slack=np.zeros((P, 10), _f), flip_budget=np.zeros((P, 10), _f))
    mask = np.zeros((H, W), np.uint8)
    pl = st["point_list"] if st["num_rendered"] > 0 else np.zeros(1, np.uint32)
    lib.gsaj_oracle_error_model(
        ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"]), _p(pl), _p(st["means2D"]),
        _p(st["conic_opacity"]), _p(st["features"]), _p(st["depths"]), _p(st["inputs"]["bg"]), _p(dLc), _p(dLd),
        ctypes.c_float(border_rel), ctypes.c_float(border_rel_T), _p(out["term_mass"]), _p(out["cond_slack"]),
        _p(out["flip_budget"]), _p(mask))
    out["border_mask"] = mask.astype(bool)
    return out


def chain(st, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, projmatrix_raw, f64=False):
    """The per-Gaussian half of the backward on its own (backward.cu:150-624): from the reverse compositor's per-Gaussian
    sums dL/dmean2D [P,3], dL/dconic [P,2,2], dL/dcolor [P,3], dL/ddepth [P,1] to dL/d{mean3D, cov3D, sh, scale, rot, tau}.
    The parity tests also feed it the DEVICE's compositor sums, to judge the device's per-Gaussian arithmetic separately
    from its summation order.  f64=True: the same operations with every intermediate in double (chain_body.inc instantiated
    in fp64) -- the yardstick that separates the rounding of an fp32 evaluation from a wrong formula."""
    lib = _load()
    P, D, M, W, H = st["P"], st["D"], st["M"], st["W"], st["H"]
    inp = st["inputs"]
    praw = _cf(projmatrix_raw).reshape(16)
    ft = np.float64 if f64 else _f
    fn = lib.gsaj_oracle_preprocess_backward_f64 if f64 else lib.gsaj_oracle_preprocess_backward
    g = dict(dL_dmean3D=np.zeros((P, 3), ft), dL_dcov3D=np.zeros((P, 6), ft), dL_dsh=np.zeros((P, M, 3), ft),
             dL_dscale=np.zeros((P, 3), ft), dL_drot=np.zeros((P, 4), ft), dL_dtau=np.zeros((P, 6), ft))
    m2, cn = _cf(dL_dmean2D).reshape(P, 3), _cf(dL_dconic).reshape(P, 4)
    dc, dd = _cf(dL_dcolor).reshape(P, 3), _cf(dL_ddepth).reshape(P)
    cov3Ds = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else st["cov3D"]
    fn(
        ctypes.c_int(P), ctypes.c_int(D), ctypes.c_int(M), ctypes.c_int(W), ctypes.c_int(H), _p(inp["means3D"]),
        _p(st["radii"]), _p(inp["shs"]), _p(st["clamped"]), _p(inp["scales"]), _p(inp["rotations"]),
        ctypes.c_float(inp["scale_modifier"]), _p(cov3Ds), _p(inp["viewmatrix"]), _p(inp["projmatrix"]), _p(praw),
        _p(inp["campos"]), ctypes.c_float(inp["tanfovx"]), ctypes.c_float(inp["tanfovy"]), _p(m2),
        _p(cn), _p(dc), _p(dd), _p(g["dL_dmean3D"]), _p(g["dL_dcov3D"]),
        _p(g["dL_dsh"]), _p(g["dL_dscale"]), _p(g["dL_drot"]), _p(g["dL_dtau"]))
    # diff_gaussian_rasterization/__init__.py:162-164: sum over Gaussians, rho = [:3], theta = [3:]
    g["dL_dtau_sum"] = g["dL_dtau"].astype(np.float64).sum(axis=0)
    return g
