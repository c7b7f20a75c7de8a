"""CPU: the loss / seed oracle against outputs of the reference's own utils/slam_utils.py functions (goldens)."""
import glob
import os

import numpy as np
import pytest

from oracle import loss_oracle as lo

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss_seed*_64x48.npz")))
KINDS = {"tracking": lo.TRACKING, "mapping": 0, "mapping_init": lo.NO_EXPOSURE}


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_loss_oracle_matches_reference_outputs(path, kind):
    g = np.load(path)
    flags = KINDS[kind] | (lo.MONOCULAR if bool(g["monocular"]) else 0)
    o = lo.loss_and_seeds(flags, g["image"], g["depth"], g["opacity"], g["gt"], g["gt_depth"], g["grad_mask"],
                          g["exposure_a"], g["exposure_b"], float(g["alpha"]), float(g["rgb_boundary_threshold"]))
    assert abs(o["loss"] - float(g[kind + "_loss"])) < 2e-7
    np.testing.assert_allclose(o["dL_dimage"], g[kind + "_dL_dimage"], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(o["dL_ddepth"], g[kind + "_dL_ddepth"], rtol=1e-5, atol=1e-10)
    if kind == "tracking":
        np.testing.assert_allclose(o["dL_dopacity"], g[kind + "_dL_dopacity"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(o["dL_da"], float(g[kind + "_dL_da"][0]), rtol=2e-4, atol=1e-8)
    np.testing.assert_allclose(o["dL_db"], float(g[kind + "_dL_db"][0]), rtol=2e-4, atol=1e-8)
