"""opt-steps/s of the Mlp DQN agent for a few network / batch sizes: one-workgroup step (LDS or global variant) vs layer-by-layer."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import border_amd as B

def run(units, bs, env):
    for k in ("BDR_NO_MLP_FUSED", "BDR_NO_MLP_LDS", "BDR_STEP_GRAPH", "BDR_NO_SMALL_GEMM", "BDR_NO_MLP_HEAD_FUSE"): os.environ.pop(k, None)
    for k in env: os.environ[k.split("=")[0]] = k.split("=")[1] if "=" in k else "1"
    rng = np.random.default_rng(0)
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=10000, seed=42), (4,), np.float32)
    n = 5000
    rb.push(rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)).astype(np.int64),
            rng.standard_normal((n, 4)).astype(np.float32), np.ones(n, np.float32), (rng.random(n) < .1).astype(np.int8), np.zeros(n, np.int8))
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=tuple(units), out_dim=2), opt_config=B.OptimizerConfig.Adam(1e-3)),
                      soft_update_interval=1, batch_size=bs, tau=0.01, critic_loss="Mse", device=0)
    a = B.Dqn.build(cfg)
    for _ in range(200): a.opt(rb)
    a.sync()
    N = 3000
    t0 = time.perf_counter()
    for _ in range(N): a.opt(rb)
    a.sync()
    dt = time.perf_counter() - t0
    a.close(); rb.close()
    return N / dt

for units, bs in (((64, 64), 32), ((256, 256), 64), ((256, 256), 128), ((128, 128), 64), ((64, 64), 128)):
    r = {name: run(units, bs, env) for name, env in (("default", ()), ("eager", ("BDR_STEP_GRAPH=0",)), ("graph", ("BDR_STEP_GRAPH=1",)), ("no_head_fuse", ("BDR_NO_MLP_HEAD_FUSE",)), ("no_head_fuse_graph", ("BDR_NO_MLP_HEAD_FUSE", "BDR_STEP_GRAPH=1")), ("layers_64x64", ("BDR_NO_MLP_FUSED", "BDR_NO_SMALL_GEMM")))}
    print(units, bs, {k: round(v) for k, v in r.items()})
