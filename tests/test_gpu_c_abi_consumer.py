"""The C ABI used from plain C (examples/c_abi_frame.c: gcc, libgsaj_hip.so + the HIP runtime, no Python, no torch in the
consumer): the drop-in boundary a maintainer of the reference binds.  CPU: the example compiles and links
against include/gsaj.h as C11.  GPU: the program renders a frame and back-propagates it from a scene file; its outputs must be
the bits the Python binding gets for the same inputs (same library, same entry points) and hold the oracle's tolerances."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import helpers as hp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.join(ROOT, "gs-slam-analytica_jacobian_amd", "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(out):
    from gsaj import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_frame.c"), "-L" + LIB_DIR, "-lgsaj_hip",
           "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_consumer_compiles_and_links_as_c11(tmp_path):
    exe = _build(str(tmp_path / "c_abi_frame"))
    # every gsaj_* symbol the program needs is resolved by the library (no compute call here: no GPU in this test)
    nm = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    wanted = {l.split()[-1].split("@")[0] for l in nm.splitlines() if "gsaj_" in l}
    assert {"gsaj_rasterize_forward", "gsaj_rasterize_backward", "gsaj_geom_workspace_bytes", "gsaj_last_error"} <= wanted
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB_DIR, "libgsaj_hip.so")], capture_output=True, text=True).stdout
    for s in wanted:
        assert (" T " + s) in exported, s


@pytest.mark.gpu
def test_c_consumer_gets_the_bits_of_the_python_binding(tmp_path):
    import torch

    exe = _build(str(tmp_path / "c_abi_frame"))
    cam, sc, deg = hp.make("p2000_160x120")
    P, M, W, H = sc["means3D"].shape[0], sc["shs"].shape[1], cam["W"], cam["H"]
    dLc, dLd = hp.seeds(cam, seed=9)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32).tobytes()  # noqa: E731
    with open(tmp_path / "scene.bin", "wb") as fh:
        fh.write(struct.pack("<5i", P, deg, M, W, H))
        fh.write(struct.pack("<5f", cam["tanfovx"], cam["tanfovy"], *bg))
        for a in (cam["viewmatrix"], cam["projmatrix"], cam["projmatrix_raw"], cam["campos"], sc["means3D"], sc["opacities"], sc["scales"],
                  sc["rotations"], sc["shs"], dLc, dLd):
            fh.write(f32(a))
    r = subprocess.run([exe, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    off = [0]

    def take(n, dt):
        a = np.frombuffer(raw, dt, n, off[0])
        off[0] += a.nbytes
        return a

    R = int(take(1, np.int32)[0])
    color, depth, opacity = take(3 * H * W, np.float32).reshape(3, H, W), take(H * W, np.float32).reshape(1, H, W), take(H * W, np.float32)
    radii, n_touched = take(P, np.int32), take(P, np.int32)
    g_mean3D, tau = take(P * 3, np.float32).reshape(P, 3), take(6, np.float32)
    assert off[0] == len(raw)
    # the Python binding on the same inputs
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw)
    g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
    got = dict(zip(hp.GRAD_NAMES, g))
    assert R == out[0] == ref["num_rendered"]
    assert np.array_equal(color, out[1].cpu().numpy()) and np.array_equal(depth, out[6].cpu().numpy())
    assert np.array_equal(radii, out[2].cpu().numpy()) and np.array_equal(n_touched, out[8].cpu().numpy())
    assert np.array_equal(g_mean3D, got["dL_dmean3D"].cpu().numpy())
    assert np.array_equal(tau, got["dL_dtau_sum"].cpu().numpy().reshape(6))
    # ... and the oracle
    hp.assert_image_close(color, ref["color"], hp.IMG_TOL, st=st, tag="c_abi/color")
    assert torch.cuda.is_available()
