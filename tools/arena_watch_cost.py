"""What the arena watch (gsaj.rasterizer.ArenaWatch: a 16-byte-per-view counter copy + an event every 4th asynchronous window) costs a
cfg2 mapping window: ms per window with auto_grow on / off, twice each.  GPU box: python tools/arena_watch_cost.py"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-slam-analytica_jacobian_amd"))
from gsaj import synthetic as syn
from gsaj.rasterizer import BatchContext
dev = torch.device("cuda:0")
cam, sc = syn.config_scene("cfg2")
K = 8
cams = syn.keyframe_cameras(K, W=cam["W"], H=cam["H"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
kw = dict(shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]), sh_degree=3)
views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
bg, means, opac, praw = torch.zeros(3, device=dev), t(sc["means3D"]), t(sc["opacities"]), t(cams[0]["projmatrix_raw"])
rng = np.random.default_rng(0)
dLc, dLd = t(rng.normal(size=(K, 3, H, W)) / (3 * H * W)), t(rng.normal(size=(K, 1, H, W)) / (H * W))
for auto in (True, False, True, False):
    bc = BatchContext(K, P, W, H, M, dev)
    bc.auto_grow = auto
    bc.forward(bg, means, opac, views, projs, cps, cam["tanfovx"], cam["tanfovy"], **kw, sync=True)
    def step():
        bc.forward(bg, means, opac, views, projs, cps, cam["tanfovx"], cam["tanfovy"], **kw, sync=False)
        bc.backward(bg, means, views, projs, praw, cps, cam["tanfovx"], cam["tanfovy"], dLc, dLd, **kw)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    print("auto_grow", auto, "ms/window %.4f" % ((time.perf_counter() - t0) / 200 * 1e3), "grown", bc.watch.grown)
