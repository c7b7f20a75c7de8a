"""CPU: the pose-step oracle against the reference's own update_pose + torch.optim.Adam (goldens)."""
import os

import numpy as np

from oracle import pose_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_adam_steps.npz"))


def test_pose_adam_oracle_matches_reference_steps():
    lr = G["lr"]
    st = po.PoseAdam(G["w2c0"], lr[0], lr[1], lr[2], lr[3], threshold=float(G["threshold"]))
    for k in range(G["g_tau"].shape[0]):
        o = st.step(G["g_tau"][k], G["g_exp"][k])
        np.testing.assert_allclose(o["tau"], G["tau"][k], rtol=2e-5, atol=1e-9, err_msg="tau step %d" % k)
        np.testing.assert_allclose(o["w2c"], G["w2c"][k], rtol=0, atol=3e-6, err_msg="w2c step %d" % k)
        np.testing.assert_allclose(o["exposure"], G["exposure"][k], rtol=2e-5, atol=1e-8)
        assert o["converged"] == bool(G["converged"][k]), k
    assert G["converged"].any() and not G["converged"].all()
    assert (np.linalg.norm(G["tau"][:, 3:], axis=1) < 1e-5).any()  # the small-angle branch is exercised
