#!/usr/bin/env python3
"""GPU box: random scenes (the fuzz cases of tests/test_gpu_random.py) through the asynchronous forward with the LDS sort sized for a
STALE, short tile-list capacity (128 .. 4096 keys), so that most tiles take sort_long_list (LDS-sized chunks + merge-path passes):
point_list / ranges bit for bit the oracle's, image bit for bit the synchronous frame's.  usage: fuzz_long_lists.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
import helpers as hp  # noqa: E402
import test_gpu_random as tr  # noqa: E402
import torch  # noqa: E402
from gsaj import rasterizer as C  # noqa: E402
from gsaj.rasterizer import FrameContext  # noqa: E402

dev = torch.device("cuda:0")
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
bad = longs = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    P, W, H, deg, cam, sc, bg3, bits = tr._fuzz_case(seed)
    rng = np.random.default_rng(seed + 7)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg3, record_bits=bits)
    M = sc["shs"].shape[1]
    a = (t(np.array(bg3)), t(sc["means3D"]), t(sc["opacities"]), t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), cam["tanfovx"], cam["tanfovy"])
    g = dict(sh_degree=deg, shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    fc = FrameContext(P, W, H, M, dev, record_bits=bits)
    R = fc.forward(*a, sync=True, **g)
    color = fc.color.clone()
    cap = int(rng.choice([100, 200, 500, 1000, 3000]))
    fc.auto_grow = False
    fc.tile_list_capacity = cap
    fc.forward(*a, sync=False, **g)
    torch.cuda.synchronize(dev)
    try:
        Rs, longest = fc.status()
        longs += longest > cap
        assert Rs == ref["num_rendered"] == R
        dbg = C.debug_export(P, fc.capacity, W, H, fc.geom, fc.binning, fc.img)
        np.testing.assert_array_equal(dbg["point_list"].cpu().numpy().astype(np.uint32)[:R], st["point_list"])
        np.testing.assert_array_equal(dbg["ranges"].cpu().numpy(), st["ranges"])
        assert torch.equal(fc.color, color), "image differs from the synchronous frame"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(seed, P, W, H, "cap", cap, type(e).__name__, str(e).replace("\n", " ")[:240])
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]), "(%d with lists longer than the capacity)" % longs)
