#!/usr/bin/env python3
"""Per-wave schedule of every kernel of a batched cfg2 window (8 keyframes): when waves start and end inside each launch.
Needs the trace build: make -C gs-slam-analytica_jacobian_amd/csrc OUT=../lib/trace EXTRA=-DGSAJ_BLOCK_TRACE, then
GSAJ_LIB_PATH=.../lib/trace/libgsaj_hip.so python tools/batch_trace.py"""
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cam, sc = syn.config_scene(wl)
    cams = syn.keyframe_cameras(K, W=cam["W"], H=cam["H"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    bc = BatchContext(K, P, W, H, M, dev)
    kw = dict(shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]), sh_degree=int(round(M ** 0.5)) - 1)
    views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
    bg, means, opac, praw = torch.zeros(3, device=dev), t(sc["means3D"]), t(sc["opacities"]), t(cams[0]["projmatrix_raw"])
    rng = np.random.default_rng(0)
    dLc, dLd = t(rng.normal(size=(K, 3, H, W)) / (3 * H * W)), t(rng.normal(size=(K, 1, H, W)) / (H * W))
    for _ in range(3):
        bc.forward(bg, means, opac, views, projs, cps, cam["tanfovx"], cam["tanfovy"], **kw)
        bc.backward(bg, means, views, projs, praw, cps, cam["tanfovx"], cam["tanfovy"], dLc, dLd, **kw)
    torch.cuda.synchronize()
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    counts = {"pre": ((P + 255) // 256) * 4 * K, "scat": ((P + 2047) // 2048) * 8 * 4 * K, "sort": tiles * 4 * K, "fwd": tiles * 4 * K, "bwd": tiles * 4 * K,
              "gath": ((P + 255) // 256) * 4 * K, "gbb": ((P + 255) // 256) * 4 * K, "gbs": (P + 63) // 64}
    for name, n in counts.items():
        fn = getattr(lib, "gsaj_trace_read_" + name, None)
        if fn is None:
            continue
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
        n = min(n, 65536)
        buf = np.zeros((n, 4), np.uint64)
        assert fn(buf.ctypes.data, n) == 0
        s, e = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
        ok = (s > 0) & (e >= s)
        s, e = s[ok], e[ok]
        t0 = s.min()
        life = (e - s) * 0.01
        busy = life.sum() / ((e.max() - t0) * 0.01)  # average number of waves in flight
        print("%-5s waves %6d span %6.1f us | life mean %5.1f p95 %5.1f max %5.1f | start p50 %5.1f p95 %5.1f max %5.1f | end p50 %5.1f p95 %5.1f p99 %5.1f | avg waves in flight %6.0f"
              % (name, len(s), (e.max() - t0) * 0.01, life.mean(), np.percentile(life, 95), life.max(), *np.percentile((s - t0) * 0.01, [50, 95, 100]),
                 *np.percentile((e - t0) * 0.01, [50, 95, 99]), busy))
        first = int(((s - t0) * 0.01 < 3.0).sum())
        print("      resident at once (waves started within the first 3 us): %d" % first)
        if name in ("pre", "scat"):  # phase stamps (pre: indexed by workgroup only -- one view's, whichever wrote last)
            nb = ((P + 255) // 256) * 4 if name == "pre" else n
            raw = np.zeros((nb, 4), np.uint64)
            fn(raw.ctypes.data, nb)
            lo32 = np.uint64(0xffffffff)
            ph = [raw[:, 2] >> np.uint64(32), raw[:, 2] & lo32, raw[:, 3] >> np.uint64(32), raw[:, 3] & lo32]
            labels = {"pre": "per-Gaussian work / histogram flush + block scan / - / -",
                      "scat": "zero LDS + loads / LDS count / reserve (returning atomics) / stores"}[name]
            print("      phases (us, mean): %s = %s" % (labels, " / ".join("%.1f" % (x.astype(np.int64).mean() * 0.01) for x in ph)))
        if name == "bwd":
            bwd_phases(lib, n)


def bwd_phases(lib, n):
    """Where a reverse-compositor wave spends its rounds (trace build): mean us per wave, wave 0 of a workgroup (it stages and
    merges) and the others."""
    fn = getattr(lib, "gsaj_trace_read_bwdph", None)
    if fn is None:
        return
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    n = min(n, 65536)
    raw = np.zeros((n, 4), np.uint64)
    assert fn(raw.ctypes.data, n) == 0
    lo32 = np.uint64(0xffffffff)
    ph = np.stack([raw[:, 0] >> np.uint64(32), raw[:, 0] & lo32, raw[:, 1] >> np.uint64(32), raw[:, 1] & lo32, raw[:, 2] >> np.uint64(32),
                   raw[:, 2] & lo32, raw[:, 3] & lo32], 1).astype(np.int64) * 0.01
    flushes, slots = ((raw[:, 3] >> np.uint64(32)) & np.uint64(0xfff)).astype(np.int64), (raw[:, 3] >> np.uint64(44)).astype(np.int64)
    print("      bwd phase-2 batches per wave %.1f, accepted entries per wave %.1f -> %.2f of 8 slots filled on average" % (flushes.mean(), slots.mean(), slots.sum() / max(1, 8 * flushes.sum())))
    names = "stage (row gather, zero acc) / barrier 1 / phases 1+2 / barrier 2 / merge + row stores / barrier 3 / loop head"
    for label, sel in (("wave 0", np.arange(n) % 4 == 0), ("waves 1-3", np.arange(n) % 4 != 0)):
        print("      bwd %s, us per wave (mean): %s = %s  (sum %.1f)" % (label, names, " / ".join("%.1f" % x for x in ph[sel].mean(0)), ph[sel].sum(1).mean()))


if __name__ == "__main__":
    main()
