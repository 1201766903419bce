"""End-to-end online loop rate (Trainer::train with the synthetic env): Policy::sample (B=1 forward + exploration),
env step on the host, push (56 KB over PCIe), Agent::opt.  Not the headline metric (that one excludes env stepping)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import border_amd as B

rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100_000, seed=42), (4, 1, 84, 84), "uint8")
rb.fill_synthetic(50_000, seed=0, kind=0, n_actions=6)
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  soft_update_interval=10000, batch_size=256, critic_loss="SmoothL1", device=0, param_seed=0)
a = B.Dqn.build(cfg)
a.set_explorer(B.EpsilonGreedy.with_final_step(1_000_000), seed=1)
env = B.SyntheticEnv((4, 1, 84, 84), np.uint8, seed=3, p_term=0.005)
for opt_interval in (1, 4):
    tr = B.Trainer(B.TrainerConfig(max_opts=300, opt_interval=opt_interval, warmup_period=0))
    t0 = time.perf_counter()
    tr.train(env, B.SimpleStepProcessor(), a, rb)
    a.sync()
    dt = time.perf_counter() - t0
    print(f"opt_interval={opt_interval}: {tr.env_steps/dt:8.1f} env steps/s, {tr.opt_steps/dt:8.1f} opt steps/s, "
          f"sample+push {1e6*tr.timer_for_samples/tr.env_steps:7.1f} us/env step")
