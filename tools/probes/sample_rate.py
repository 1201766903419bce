"""Policy::sample latency / throughput vs n_procs (Nature-CNN DQN, eps-greedy)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import border_amd as B
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  batch_size=256, device=0, train=True)
a = B.Dqn.build(cfg)
a.set_explorer(B.EpsilonGreedy.with_final_step(1_000_000), seed=1)
rng = np.random.default_rng(0)
for n in (1, 8, 64, 256, 1024):
    obs = rng.integers(0, 256, (n, 4, 1, 84, 84), dtype=np.uint8)
    for _ in range(20):
        a.sample(obs)
    t0 = time.perf_counter()
    it = 200 if n <= 64 else 50
    for _ in range(it):
        a.sample(obs)
    dt = (time.perf_counter() - t0) / it
    print(f"n_procs={n:5d}: {1e6*dt:8.1f} us per call, {n/dt:10.0f} actions/s")
