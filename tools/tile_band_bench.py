#!/usr/bin/env python3
"""Tile-band sharding of one tracking iteration (gsaj.tile_band_shard), per-rank time measured on ONE GPU.

A tracking iteration = asynchronous forward -> tracking loss seeds -> backward(pose_only) -> device Adam / update_pose.
For world = 1, 2, 4, 8 the frame is cut into bands balanced by the Gaussian-pixel interactions of a whole-frame probe, each
band is timed on its own (what that rank would run), and the slowest band is the iteration time of the sharded loop
WITHOUT its one collective (an 11-float all-reduce; needs the 8-GPU node the driver has).  Prints one JSON object.
usage: tile_band_bench.py [workload=cfg2] [iterations=200]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
import torch  # noqa: E402
from gsaj import synthetic as syn, tile_band_shard as tbs  # noqa: E402
from gsaj import rasterizer as C  # noqa: E402
from gsaj.losses import LossSeeds, TRACKING  # noqa: E402
from gsaj.pose_step import PoseTracker  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    cam, sc = syn.config_scene(wl)
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    deg = int(round(M ** 0.5)) - 1
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    means, opac = t(sc["means3D"]), t(sc["opacities"])
    kw = dict(shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]), sh_degree=deg)
    bg, praw = torch.zeros(3, device=dev), t(cam["projmatrix_raw"])
    rng = np.random.default_rng(0)
    gt_color, gt_depth = t(rng.uniform(0, 1, (3, H, W))), t(rng.uniform(0.5, 4, (H, W)))
    w2c = torch.as_tensor(np.ascontiguousarray(cam["viewmatrix"].T), dtype=torch.float32)

    use_graph = os.environ.get("GSAJ_BAND_GRAPH", "1") != "0"

    def iteration_ms(band):
        pose = PoseTracker(w2c, praw, dev, lr_rot=0.0, lr_trans=0.0)  # learning rate 0: every iteration renders the same frame
        fc, ls = C.FrameContext(P, W, H, M, dev), LossSeeds(W, H, dev)
        if band is not None:
            fc.set_tile_band(*band)
        packed = torch.zeros(tbs.REDUCED_FLOATS, device=dev)

        def it(sync):
            fc.forward(bg, means, opac, pose.viewmatrix, pose.projmatrix, pose.campos, cam["tanfovx"], cam["tanfovy"], sync=sync, **kw)
            L = ls(TRACKING, 0.9, 0.01, fc.color, fc.depth, fc.opacity, gt_color, gt_depth, None, pose.exposure_a, pose.exposure_b)
            g = fc.backward(bg, means, pose.viewmatrix, pose.projmatrix, praw, pose.campos, cam["tanfovx"], cam["tanfovy"],
                            L["dL_dcolor"], L["dL_ddepth"], pose_only=True, **kw)
            tbs.pack_pose_terms(g["tau_sum"], ls.scalars, out=packed)
            pose.step(packed[tbs.TAU], packed[tbs.EXPOSURE_GRADS])

        it(True)
        for _ in range(10):
            it(False)
        torch.cuda.synchronize()
        run = lambda: it(False)  # noqa: E731
        if use_graph:  # the iteration has no host round trip: capture its ~14 launches once, replay them as one hipGraph
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                it(False)
            run = graph.replay
            for _ in range(3):
                run()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        fc.status()
        if os.environ.get("GSAJ_BAND_STAGES"):
            with C.profile_stages(max_records=20 * 16) as prof:
                for _ in range(20):
                    it(False)
                torch.cuda.synchronize()
            print(band, {k: round(1e3 * v / 20, 1) for k, v in prof.ms.items() if v > 0}, file=sys.stderr)
        return ms, fc

    whole_ms, probe = iteration_ms(None)
    dbg = C.debug_export(P, probe.R, W, H, probe.geom, probe.binning, probe.img)
    work = tbs.row_work(dbg["n_contrib"])
    out = {"workload": wl, "P": P, "W": W, "H": H, "iterations": n, "hip_graph": use_graph, "interactions": int(sum(work)),
           "whole_frame_ms_per_iteration": round(whole_ms, 4), "worlds": {}}
    for world in (2, 4, 8):
        for kind, bands in (("balanced", tbs.balanced_bands(work, world)), ("uniform", tbs.uniform_bands(H, world))):
            per = [round(iteration_ms(b)[0], 4) for b in bands]
            out["worlds"]["%d_%s" % (world, kind)] = {
                "bands": bands, "work_share": [round(sum(work[b:e]) / sum(work), 3) for b, e in bands], "ms_per_rank": per,
                "slowest_ms": max(per), "speedup_without_collective": round(whole_ms / max(per), 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
