#!/usr/bin/env python3
"""Golden for the densification bookkeeping (build container only): runs the reference's OWN
GaussianModel.add_densification_stats (gaussian_splatting/scene/gaussian_model.py:767-771; the method's `def` is taken from
the parsed file and bound to a bare object holding CPU tensors -- the module itself needs open3d / plyfile / simple_knn to
import) and the max_radii2D / n_obs lines of the mapping loop (utils/slam_backend.py:113-121, 236-250, restated: they are
statements inside a method that needs the whole SLAM back end) over a window of K views.  Stores inputs and outputs only.

    python tests/golden/make_densify_goldens.py   ->  tests/golden/densify_K4_P300.npz
"""
import ast
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gaussian_splatting/scene/gaussian_model.py"


def reference_method(name):
    tree = ast.parse(open(REF).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns[name]


def main():
    add_stats = reference_method("add_densification_stats")
    rng = np.random.default_rng(17)
    K, P = 4, 300

    class Bare:
        pass

    m = Bare()
    m.xyz_gradient_accum = torch.tensor(rng.uniform(0, 1, (P, 1)), dtype=torch.float32)
    m.denom = torch.tensor(rng.integers(0, 5, (P, 1)), dtype=torch.float32)
    m.max_radii2D = torch.tensor(rng.integers(0, 30, (P,)), dtype=torch.float32)
    init = dict(accum0=m.xyz_gradient_accum.numpy().copy(), denom0=m.denom.numpy().copy(), maxr0=m.max_radii2D.numpy().copy())
    grads = rng.normal(size=(K, P, 3)).astype(np.float32)
    radii = (rng.integers(0, 40, (K, P)) * (rng.uniform(size=(K, P)) > 0.35)).astype(np.int32)
    n_touched = (rng.integers(0, 9, (K, P)) * (rng.uniform(size=(K, P)) > 0.5)).astype(np.int32)
    n_obs = torch.zeros(P).int()
    for k in range(K):
        vs = Bare()
        vs.grad = torch.tensor(grads[k])
        vis = torch.tensor(radii[k] > 0)
        r = torch.tensor(radii[k]).float()
        m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], r[vis])       # slam_backend.py:115-118
        add_stats(m, vs, vis)                                             # gaussian_model.py:767-771
        n_obs += (torch.tensor(n_touched[k]) > 0).long().int()            # slam_backend.py:240, 248-250
    np.savez_compressed(os.path.join(HERE, "densify_K4_P300.npz"), grads=grads, radii=radii, n_touched=n_touched, **init,
                        accum=m.xyz_gradient_accum.numpy(), denom=m.denom.numpy(), maxr=m.max_radii2D.numpy(), n_obs=n_obs.numpy())
    print("ok", os.path.getsize(os.path.join(HERE, "densify_K4_P300.npz")))


if __name__ == "__main__":
    main()
