import time, sys
sys.path.insert(0, '/root/repo')
import torch
import border_amd as B
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100000, seed=42), (4, 1, 84, 84), "uint8", device=0)
rb.fill_synthetic(100000, seed=0, kind=0, n_actions=6)
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  soft_update_interval=10000, batch_size=256, critic_loss="SmoothL1", device=0, param_seed=0)
a = B.Dqn.build(cfg); a.train()
for _ in range(50): a.opt(rb)
a.sync()
for n in (4, 8, 16, 32, 64):
    a.sync(); time.sleep(0.01)
    t0 = time.perf_counter()
    for _ in range(n): a.opt(rb)
    t1 = time.perf_counter()
    a.sync()
    t2 = time.perf_counter()
    print(f"n={n:3d}: issue {1e6*(t1-t0)/n:7.1f} us/opt   total {1e6*(t2-t0)/n:7.1f} us/opt")
