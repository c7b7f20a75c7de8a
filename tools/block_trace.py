#!/usr/bin/env python3
"""Per-wave schedule trace of k_render_fwd / k_render_bwd on the benchmark workload.

Needs a library built with the trace hooks:   make -C gs-slam-analytica_jacobian_amd/csrc clean all EXTRA=-DGSAJ_BLOCK_TRACE
Writes gpurun_out/block_trace_{fwd,bwd}.npy  ([waves, 4] uint64: start, end (10 ns ticks), HW_ID, XCC_ID) and prints a summary.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
import torch  # noqa: E402
from gsaj import _lib, synthetic as syn  # noqa: E402
from gsaj.rasterizer import FrameContext  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cam, sc = syn.config_scene(wl)
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    ctx = FrameContext(P, W, H, M, dev)
    fa = dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), viewmatrix=t(cam["viewmatrix"]),
              projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3,
              shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    rng = np.random.default_rng(0)
    dLc, dLd = t(rng.normal(size=(3, H, W)) / (3 * H * W)), t(rng.normal(size=(1, H, W)) / (H * W))
    for _ in range(3):
        ctx.forward(**fa)
        ctx.backward(bg=fa["bg"], means3D=fa["means3D"], viewmatrix=fa["viewmatrix"], projmatrix=fa["projmatrix"],
                     projmatrix_raw=t(cam["projmatrix_raw"]), campos=fa["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                     dL_dcolor=dLc, dL_ddepth=dLd, sh_degree=3, shs=fa["shs"], scales=fa["scales"], rotations=fa["rotations"])
    torch.cuda.synchronize()
    nw = ((W + 15) // 16) * ((H + 15) // 16) * 4
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for name in ("pre", "scat", "fwd", "bwd", "sort", "gbwd"):
        if not hasattr(lib, "gsaj_trace_read_" + name):
            continue
        fn = getattr(lib, "gsaj_trace_read_" + name)
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
        n_rec = (P + 63) // 64 if name == "gbwd" else (((P + 255) // 256) * 4 if name in ("pre", "scat") else nw)
        buf = np.zeros((n_rec, 4), np.uint64)
        assert fn(buf.ctypes.data, n_rec) == 0
        np.save(os.path.join(ROOT, "gpurun_out", "block_trace_%s.npy" % name), buf)
        s, e = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
        t0 = s.min()
        life = (e - s) * 0.01
        print(name, "waves", len(buf), "span us %.1f" % ((e.max() - t0) * 0.01), "mean life us %.1f" % life.mean(),
              "p5/p50/p95 life %.1f %.1f %.1f" % tuple(np.percentile(life, [5, 50, 95])),
              "start p50/p95/max us %.1f %.1f %.1f" % tuple(np.percentile((s - t0) * 0.01, [50, 95, 100])),
              "end p5/p50/p95 us %.1f %.1f %.1f" % tuple(np.percentile((e - t0) * 0.01, [5, 50, 95])))
        if name == "pre":
            lo32 = np.uint64(0xffffffff)
            ph = [buf[:, 2] >> np.uint64(32), buf[:, 2] & lo32, buf[:, 3] >> np.uint64(32), buf[:, 3] & lo32]
            print("   preprocess phases per wave (us): per-Gaussian work %.1f  histogram flush + block scan %.1f  drain + ticket %.1f  "
                  "frame scan (last workgroup only; mean / max) %.1f / %.1f" % tuple(
                      [x.astype(np.int64).mean() * 0.01 for x in ph] + [ph[3].astype(np.int64).max() * 0.01]))
        if name == "scat":
            lo32 = np.uint64(0xffffffff)
            ph = [buf[:, 2] >> np.uint64(32), buf[:, 2] & lo32, buf[:, 3] >> np.uint64(32), buf[:, 3] & lo32]
            print("   scatter phases per wave (us): zero LDS + loads %.1f  LDS count %.1f  reserve (returning atomics) %.1f  stores %.1f" % tuple(
                x.astype(np.int64).mean() * 0.01 for x in ph))
        if name == "gbwd":
            m21 = np.uint64(0x1fffff)
            a, b = buf[:, 2], buf[:, 3]
            ph = [(a >> np.uint64(42)), (a >> np.uint64(21)) & m21, a & m21, (b >> np.uint64(21)) & m21, b & m21]
            print("   gbwd phases per wave (us): inputs+SH stage %.1f  gather rows %.1f  math+SH %.1f  tau hand-off %.1f  stores %.1f" % tuple(
                x.astype(np.int64).mean() * 0.01 for x in ph))
        if name == "sort":
            print("   sort phases per wave (us): load keys %.1f  sort %.1f  ids+records %.1f" % (
                buf[:, 2].astype(np.int64).mean() * 0.01, buf[:, 3].astype(np.int64).mean() * 0.01,
                (life - (buf[:, 2] + buf[:, 3]).astype(np.int64) * 0.01).mean()))


if __name__ == "__main__":
    main()
