"""ctypes binding of libgsaj_hip.so (C ABI: include/gsaj.h).

The library is the product: there is no Python / CPU fallback.  If it has not been built
(`python __graft_entry__.py build` or `make -C gs-slam-analytica_jacobian_amd/csrc`) importing
this module raises ImportError -- loudly, so a GPU run can never silently use something else.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.environ.get("GSAJ_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libgsaj_hip.so")  # override: A/B kernel experiments
HEADER = os.path.join(os.path.dirname(PKG_ROOT), "include", "gsaj.h")

c_int, c_float, c_double, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p
P = c_void_p  # every device / host pointer travels as void*

# name -> (restype, argtypes); must list every function include/gsaj.h declares
SIGNATURES = {
    "gsaj_last_error": (ctypes.c_char_p, []),
    "gsaj_version": (c_int, []),
    "gsaj_geom_workspace_bytes": (c_size_t, [c_int]),
    "gsaj_image_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gsaj_binning_workspace_bytes": (c_size_t, [c_int]),
    "gsaj_forward_preprocess": (c_int, [c_int] * 5 + [P, P, P, P, P, c_float, P, P, P, P, P, c_float, c_float, c_int, P, P, P, P, P]),
    "gsaj_forward_num_rendered": (c_int, [c_int, c_int, P, P, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gsaj_forward_render": (c_int, [c_int] * 5 + [P, P, P, P, P, c_size_t, P, P, P, P, P, c_int, P]),
    "gsaj_rasterize_forward": (c_int, [c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, c_float, P, P, P, P, P,
                                       c_float, c_float, c_int, P, P, P, P, P, P, P, c_size_t, P,
                                       ctypes.POINTER(c_int), c_int, P]),
    "gsaj_rasterize_forward_async": (c_int, [c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, c_float, P, P, P, P, P,
                                             c_float, c_float, c_int, P, P, P, P, P, P, P, c_size_t, c_int, c_int, P, c_int, P]),
    "gsaj_rasterize_forward_batch": (c_int, [c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, c_float, P, P, P, P, P,
                                             c_float, c_float, c_int, P, P, P, P, P, P, P, c_size_t, c_int, c_int, P, c_int, P]),
    "gsaj_rasterize_backward_batch": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, c_float, P, P, P, P,
                                              P, P, c_float, c_float, P, P, P, P, P, P] + [P] * 12 + [c_int, P]),
    "gsaj_fused_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gsaj_rasterize_forward_loss": (c_int, [c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, c_float, P, P, P, P, P,
                                            c_float, c_float, c_int, P, P, P, P, P, P, P, c_size_t, c_int, c_int, P, c_int,
                                            c_int, c_float, c_float, P, P, P, P, P, P, P, P]),
    "gsaj_rasterize_backward_loss": (c_int, [c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, c_float, P, P, P, P, P, P,
                                             c_float, c_float, P, P, P, P, c_int, c_float, c_float, P, P, P, P, P, P, P, P] + [P] * 12 + [P]),
    "gsaj_forward_aborted_count": (c_int, [c_int, c_int, P, P, ctypes.POINTER(c_int)]),
    "gsaj_set_tile_band": (c_int, [c_int, c_int, P, c_int, c_int, P]),
    "gsaj_forward_preprocess_cap": (c_int, [c_int] * 5 + [P, P, P, P, P, c_float, P, P, P, P, P, c_float, c_float, c_int, P, P,
                                            P, P, c_int, c_int, P]),
    "gsaj_rasterize_backward": (c_int, [c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, c_float, P, P, P, P, P, P,
                                        c_float, c_float, P, P, P, P, P, P] + [P] * 12 + [P]),
    "gsaj_mark_visible": (c_int, [c_int, P, P, P, P, P]),
    "gsaj_debug_export": (c_int, [c_int] * 4 + [P] * 3 + [P] * 11 + [P]),
    "gsaj_debug_export_view_sums": (c_int, [c_int, P, P, P]),
    "gsaj_profile_begin": (c_int, [c_int]),
    "gsaj_profile_end": (c_int, [P, P]),
    "gsaj_dense_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gsaj_dense_backward": (c_int, [c_int, c_int, c_int] + [P] * 7 + [P] * 4 + [P, c_int, P, P]),
    "gsaj_dense_project_workspace_bytes": (c_size_t, [c_int]),
    "gsaj_dense_project": (c_int, [c_int] * 5 + [P] * 6 + [c_double, c_double] + [P] * 6 + [P, P]),
    "gsaj_dense_render": (c_int, [c_int, c_int, c_int] + [P] * 5 + [P, P, P]),
    "gsaj_pose_jacobians": (c_int, [c_int, P, P, P, c_double, c_double, c_int, c_int, P, P, P]),
    "gsaj_dense_tau": (c_int, [c_int, c_int, c_int] + [P] * 11 + [P, P, P]),
    "gsaj_dist2_workspace_bytes": (c_size_t, [c_int]),
    "gsaj_dist2": (c_int, [c_int, P, P, P, P]),
    "gsaj_debug_dist2_order": (c_int, [c_int, P, P, P, P]),
    "gsaj_pose_state_floats": (c_int, []),
    "gsaj_pose_adam_step": (c_int, [P, P] + [c_float] * 8 + [P, P, P, P]),
    "gsaj_pose_adam_step_batch": (c_int, [c_int, P, P, P] + [c_float] * 8 + [P, P, P, c_size_t, P]),
    "gsaj_forward_abort_flag": (c_void_p, [c_int, c_int, P]),
    "gsaj_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gsaj_densification_stats": (c_int, [c_int, c_int] + [P] * 7 + [P]),
    "gsaj_isotropic_workspace_bytes": (c_size_t, [c_int]),
    "gsaj_isotropic_loss": (c_int, [c_int, c_int, c_float, P, P, c_int, P, P, P]),
    "gsaj_loss_seeds": (c_int, [c_int, c_int, c_int, c_float, c_float] + [P] * 8 + [P] * 4 + [P, P]),
    "gsaj_loss_seeds_batch": (c_int, [c_int] + [c_int, c_int, c_int, c_float, c_float] + [P] * 8 + [P] * 4 + [P, P]),
}

_lib = None


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of every kernel file into lib/libgsaj_hip.so."""
    cmd = ["make", "-C", CSRC, "-j8"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libgsaj_hip.so is missing (%s). Build the HIP extension first: "
            "`python __graft_entry__.py` or `make -C %s`. There is no CPU fallback." % (LIB_PATH, CSRC))
    # PyTorch supplies device memory and streams, so ITS HIP runtime must be the one the process
    # initialises: import it before dlopen()ing the kernels (two runtimes in one process fail with
    # "no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class GsajError(RuntimeError):
    pass


def check(rc, what):
    """Negative return codes become exceptions carrying gsaj_last_error()."""
    if rc < 0:
        msg = load().gsaj_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise Exception(msg or ("%s: invalid argument" % what))  # reference raises plain Exception for bad combos
        raise GsajError("%s failed (%d): %s" % (what, rc, msg))
    return rc
