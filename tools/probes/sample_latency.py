"""Latency of Policy::sample (bdr_agent_sample: host obs -> device forward -> host action) per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import border_amd as B

def t(fn, n=2000):
    for _ in range(100): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6

rng = np.random.default_rng(0)
for units in ((64, 64), (256, 256)):
    a = B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=units, out_dim=2), opt_config=B.OptimizerConfig.Adam(1e-3)), device=0, batch_size=32))
    a.eval()
    for n in (1, 64):
        obs = rng.standard_normal((n, 4)).astype(np.float32)
        print(f"dqn mlp {units} n={n}: sample {t(lambda: a.sample(obs)):.1f} us  qvalues {t(lambda: a.qvalues(obs)):.1f} us")
    a.close()
a = B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)), device=0, batch_size=32))
a.eval()
for n in (1, 16):
    obs = rng.integers(0, 255, (n, 4, 1, 84, 84)).astype(np.uint8)
    print(f"dqn cnn n={n}: sample {t(lambda: a.sample(obs), 500):.1f} us")
a.close()
s = B.Sac.build(B.SacConfig(obs_dim=17, act_dim=6, pi_units=(256, 256), q_units=(256, 256), n_critics=2, batch_size=256, device=0))
s.eval()
for n in (1, 64):
    obs = rng.standard_normal((n, 17)).astype(np.float32)
    print(f"sac n={n}: sample {t(lambda: s.sample(obs)):.1f} us")
s.close()
