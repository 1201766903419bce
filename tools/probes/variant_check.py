"""Same seeded opt stream on whichever library BORDER_AMD_LIB names: prints a fingerprint of the parameters after 12 opt steps
(A/B runs compare it across kernel variants: same k order -> same bits, a different split -> ~1e-7 relative)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import border_amd as B

rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=4000, seed=42), (4, 1, 84, 84), np.uint8)
rb.fill_synthetic(4000, seed=3, kind=0, n_actions=6)
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  batch_size=256, critic_loss="SmoothL1", tau=1.0, soft_update_interval=5, device=0, param_seed=9)
a = B.Dqn.build(cfg)
for _ in range(12):
    a.opt(rb)
a.sync()
p = a.get_params("qnet").astype(np.float64)
g = a.get_params("grad").astype(np.float64)
print("fingerprint: sum|p| %.10e  sum|g| %.10e  p[::100003] %s" % (np.abs(p).sum(), np.abs(g).sum(), np.array2string(p[::400003], precision=9)))
a.close(); rb.close()
