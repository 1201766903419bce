"""IQN acting kernels (act_small.hpp through bdr_iqn_qvalues) against the training forward of the same rows."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import border_amd as B
rng = np.random.default_rng(0)
for A in (6, 18):
    a = B.Iqn.build(B.IqnConfig(n_actions=A, device=0, batch_size=32, seed=3)); a.eval()
    for n in (1, 3, 8):
        obs = rng.integers(0, 255, (n, 4, 1, 84, 84)).astype(np.uint8)
        q = a.qvalues(obs)
        os.environ["BDR_X"] = "1"
        big = np.concatenate([obs, rng.integers(0, 255, (12, 4, 1, 84, 84)).astype(np.uint8)])
        qb = a.qvalues(big)[:n]
        err = np.abs(q - qb).max() / np.abs(qb).max()
        print(f"A={A} n={n}: rel diff vs the training kernels {err:.2e}", "OK" if err < 2e-6 else "BAD")
    a.close()
