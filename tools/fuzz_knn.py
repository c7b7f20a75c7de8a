#!/usr/bin/env python3
"""GPU box: distCUDA2 (gsaj_dist2) on random point sets -- size, spread, clusters, exact duplicates, planes -- against the brute-force
oracle; and the Morton order against a stable sort of the codes.  usage: fuzz_knn.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
import torch  # noqa: E402
from oracle import knn_oracle  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([4, 5, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4097, 6000]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        p = rng.uniform(-1, 1, (n, 3)) * float(10 ** rng.uniform(-3, 3))
    elif kind == 1:
        p = rng.normal(size=(n, 3)) * float(10 ** rng.uniform(-4, -1)) + rng.integers(0, 7, (n, 1)) * 1.0
    elif kind == 2:
        p = np.concatenate([rng.uniform(-3, 3, (n, 2)), np.full((n, 1), float(rng.uniform(-5, 5)))], axis=1)
    elif kind == 3:
        k = max(1, n // int(rng.integers(2, 6)))
        p = rng.uniform(-1, 1, (k, 3))[rng.integers(0, k, n)]
    else:
        p = rng.uniform(-1, 1, (n, 3)) * np.array([1.0, 1e-3, 1e-6])[None, :]
    p = p.astype(np.float32)
    got = distCUDA2(torch.as_tensor(p, device="cuda:0")).cpu().numpy()
    want = knn_oracle.dist2(p)
    if not np.allclose(got, want, rtol=2e-5, atol=1e-12 * float(np.abs(p).max()) ** 2 + 1e-30):
        bad += 1
        i = int(np.argmax(np.abs(got - want) / (np.abs(want) + 1e-30)))
        print(seed, n, kind, "worst", i, got[i], want[i])
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]))
