#!/usr/bin/env python3
"""Per-wave phase trace of k_gaussian_bwd_batch on a cfg2 window of 8 keyframes (library built with -DGSAJ_BLOCK_TRACE:
make -C gs-slam-analytica_jacobian_amd/csrc OUT=../lib/trace EXTRA=-DGSAJ_BLOCK_TRACE; run with GSAJ_LIB_PATH=.../lib/trace/libgsaj_hip.so)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
import torch  # noqa: E402
from gsaj import _lib, synthetic as syn  # noqa: E402
from gsaj.rasterizer import BatchContext  # noqa: E402


def main():
    K = 8
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cam, sc = syn.config_scene("cfg2")
    cams = syn.keyframe_cameras(K, W=cam["W"], H=cam["H"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    bc = BatchContext(K, P, W, H, M, dev)
    kw = dict(shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]), sh_degree=3)
    views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
    bg, means, opac, praw = torch.zeros(3, device=dev), t(sc["means3D"]), t(sc["opacities"]), t(cams[0]["projmatrix_raw"])
    rng = np.random.default_rng(0)
    dLc, dLd = t(rng.normal(size=(K, 3, H, W)) / (3 * H * W)), t(rng.normal(size=(K, 1, H, W)) / (H * W))
    for _ in range(3):
        bc.forward(bg, means, opac, views, projs, cps, cam["tanfovx"], cam["tanfovy"], **kw)
        bc.backward(bg, means, views, projs, praw, cps, cam["tanfovx"], cam["tanfovy"], dLc, dLd, **kw)
    torch.cuda.synchronize()
    fn = lib.gsaj_trace_read_gbb
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    n = ((P + 63) // 64) * 4
    buf = np.zeros((n, 4), np.uint64)
    assert fn(buf.ctypes.data, n) == 0
    s, e = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
    life = (e - s) * 0.01
    print("waves", n, "span us %.1f" % ((e.max() - s.min()) * 0.01), "life mean/p50/p95 %.1f %.1f %.1f" % (life.mean(), *np.percentile(life, [50, 95])),
          "start p50/p95/max %.1f %.1f %.1f" % tuple(np.percentile((s - s.min()) * 0.01, [50, 95, 100])))
    print("   end p50/p95/p99/max %.1f %.1f %.1f %.1f; life max %.1f; starts histogram (10 us bins):" % (*np.percentile((e - s.min()) * 0.01, [50, 95, 99, 100]), life.max()),
          np.histogram((s - s.min()) * 0.01, bins=np.arange(0, 100, 10))[0].tolist())
    late = np.argsort(e)[-4:]
    print("   last waves to end: wave index", late.tolist(), "start", ((s[late] - s.min()) * 0.01).round(1).tolist(), "life", life[late].round(1).tolist())
    m21 = np.uint64(0x1fffff)
    a, b = buf[:, 2], buf[:, 3]
    ph = [(a >> np.uint64(42)), (a >> np.uint64(21)) & m21, a & m21, (b >> np.uint64(42)), (b >> np.uint64(21)) & m21, b & m21]
    names = ["inputs + SH staging + barrier", "view loads (2 views)", "chain + per-view stores + tau butterfly (2 views)", "LDS sums", "barrier + cross-wave sum",
             "ticket"]
    for nm, x in zip(names, ph):
        x = x.astype(np.int64) * 0.01
        print("   %-50s mean %.1f  p95 %.1f us" % (nm, x.mean(), np.percentile(x, 95)))
    print("   phases of the last wave to end (us):", [round(float(x[late[-1]]) * 0.01, 1) for x in ph])
    print("   stores + exit (life - phases)                      mean %.1f us" % (life - sum(x.astype(np.int64) * 0.01 for x in ph)).mean())


if __name__ == "__main__":
    main()
