"""CPU oracle (test infrastructure only) of simple_knn distCUDA2 (reference submodules/simple-knn/simple_knn.cu:149-185):
mean of the squared distances to the 3 nearest OTHER points; missing neighbours contribute FLT_MAX (-> inf in fp32).
The reference holds no fixture for this function and its CUDA source cannot run here: parity against reference OUTPUTS is
unpinned; the specification (an exact 3-NN search) is what is checked, by brute force.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np


def dist2(points):
    p = np.asarray(points, np.float32)
    n = p.shape[0]
    out = np.empty(n, np.float32)
    fmax = np.float32(np.finfo(np.float32).max)
    for s in range(0, n, 512):
        q = p[s:s + 512]
        d = ((q[:, None, :] - p[None, :, :]) ** 2).astype(np.float32).sum(axis=2, dtype=np.float32)
        d[np.arange(q.shape[0]), s + np.arange(q.shape[0])] = np.inf      # not itself (index, not value: duplicates count)
        k = min(3, n - 1)
        best = np.sort(np.partition(d, k - 1, axis=1)[:, :k], axis=1) if k > 0 else np.empty((q.shape[0], 0), np.float32)
        pad = np.full((q.shape[0], 3 - k), fmax, np.float32)
        with np.errstate(over="ignore"):
            out[s:s + 512] = (np.concatenate([best, pad], axis=1).astype(np.float32).sum(axis=1, dtype=np.float32) / np.float32(3.0))
    return out
