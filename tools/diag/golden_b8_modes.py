"""diagnostic (round 6): the step-1 gradient sample of the B = 8 double-DQN golden under both arithmetics, and where the largest deviation sits"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import border_amd as B
from oracle import torch_ref as T
g = np.load(os.path.join(ROOT, "tests", "golden", "dqn_cnn_b8_mse_ddqn.npz"))
shapes = T.cnn_shapes(6)
for mode in ("bf16x3_6", "f32_exact"):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                      device=0, batch_size=8, critic_loss="Mse", double_dqn=True, tau=0.005, soft_update_interval=1, arithmetic=mode)
    a = B.Dqn.build(cfg)
    p0 = T.init_params(shapes, 2)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    for s in range(2):
        a.update_on_batch(*T.synthetic_atari_batch(8, 6, 200 + s))
        grads = a.get_params("grad")
        st = max(1, grads.size // 4096) | 1
        d = np.abs(grads[::st].astype(np.float64) - g[f"s{s}_grads_sample"]) / np.abs(g[f"s{s}_grads_sample"]).max()
        i = int(d.argmax())
        o, var = 0, None
        for k, sh in enumerate(shapes):
            n = int(np.prod(sh))
            if o <= i * st < o + n: var = (k, sh)
            o += n
        print(mode, "step", s, "max rel", d.max(), "at sample", i, "variable", var, "n > 1e-4:", int((d > 1e-4).sum()), flush=True)
    a.close()
