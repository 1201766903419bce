"""opt-steps/s of the non-headline BASELINE configs (C1 CartPole DQN, C4 IQN, C5 SAC); synthetic rings."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import border_amd as B


def rate(agent, rb, steps, warm):
    for _ in range(warm):
        agent.opt(rb)
    agent.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.opt(rb)
    agent.sync()
    return steps / (time.perf_counter() - t0)


# C1: CartPole-shaped DQN, Mlp[64,64], replay 10k, batch 32
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=10_000, seed=42), (4,), np.float32)
rb.fill_synthetic(10_000, seed=0, kind=1, n_actions=2)
a = B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2),
                                                          opt_config=B.OptimizerConfig.Adam(1e-3)),
                            device=0, batch_size=32, tau=0.01, soft_update_interval=1, critic_loss="Mse"))
a.train()
print(f"C1 DQN CartPole Mlp[64,64] B=32      : {rate(a, rb, 3000, 200):9.0f} opt-steps/s")
a.close(); rb.close()

# C4: IQN, Nature-CNN trunk, 64 quantiles, batch 512
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=200_000, seed=42), (4, 1, 84, 84), "uint8")
rb.fill_synthetic(200_000, seed=0, kind=0, n_actions=6)
iq = B.Iqn(B.IqnConfig(n_actions=6, batch_size=512, sample_percents_pred="Uniform64", sample_percents_tgt="Uniform64", device=0, train=True))
print(f"C4 IQN Atari, 64 quantiles, B=512     : {rate(iq, rb, 100, 10):9.1f} opt-steps/s")
iq.close(); rb.close()

# C5: SAC obs 17 / act 6, twin-Q, hidden [256,256], batch 1024
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=1_000_000, seed=42), (17,), np.float32, act_shape=(6,), act_dtype=np.float32)
rb.fill_synthetic(1_000_000, seed=0, kind=1, n_actions=0)
s = B.Sac(B.SacConfig(obs_dim=17, act_dim=6, pi_units=(256, 256), q_units=(256, 256), n_critics=2, batch_size=1024,
                      ent_coef_mode=("Auto", -6.0, 3e-4), device=0, train=True))
print(f"C5 SAC 17/6 twin-Q [256,256] B=1024   : {rate(s, rb, 2000, 100):9.0f} opt-steps/s")
s.close(); rb.close()
