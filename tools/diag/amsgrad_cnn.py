"""diagnostic (round 6): where the first AdamW{amsgrad} step of the Nature-CNN agent departs from the ATen restatement, per variable"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import border_amd as B
from oracle import torch_ref as T
kw = dict(beta1=0.8, beta2=0.9, wd=0.05, eps=1e-6, amsgrad=True)
shapes, lr = T.cnn_shapes(6), 1e-4
p0 = T.init_params(shapes, 5)
batches = [T.synthetic_atari_batch(8, 6, 70 + s) for s in range(2)]
batches = [(o, a_, n, (r * 10.0).astype(np.float32), t) for (o, a_, n, r, t) in batches]
for mode in ("bf16x3_6", "f32_exact"):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.AdamW(lr, **kw)),
                      device=0, batch_size=8, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, arithmetic=mode)
    a = B.Dqn.build(cfg)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    t = T.TorchDqn("cnn", shapes, p0, lr=lr, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, adamw=kw)
    for s, batch in enumerate(batches):
        r = t.update(*batch)
        a.update_on_batch(*batch)
        g, gr = a.get_params("grad").astype(np.float64), r["grads"].astype(np.float64)
        p, pr = a.get_params("qnet").astype(np.float64), t.params()
        o = 0
        for k, sh in enumerate(shapes):
            n = int(np.prod(sh))
            dg, dp = np.abs(g[o:o + n] - gr[o:o + n]), np.abs(p[o:o + n] - pr[o:o + n])
            i = int(dp.argmax())
            print(mode, "step", s, sh, "max|g| %.3e  max dg %.3e  max dp/lr %.3f  at: g_ref %.3e g %.3e  p0 %.6f p_ref %.8f p %.8f" %
                  (np.abs(gr[o:o + n]).max(), dg.max(), dp.max() / lr, gr[o + i], g[o + i], p0[o + i], pr[o + i], p[o + i]), flush=True)
            o += n
    a.close()
