"""Host enqueue time vs device time of the SAC opt step, graph replay vs eager launches (BDR_NO_STEP_GRAPH=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import border_amd as B

od, ad, Bsz = 17, 6, 1024
rng = np.random.default_rng(0)
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100000, seed=42), (od,), np.float32, (ad,), np.float32)
n = 50000
rb.push(rng.standard_normal((n, od)).astype(np.float32), rng.uniform(-1, 1, (n, ad)).astype(np.float32),
        rng.standard_normal((n, od)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
        (rng.random(n) < .05).astype(np.int8), np.zeros(n, np.int8))
cfg = B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(256, 256), q_units=(256, 256), n_critics=2, batch_size=Bsz,
                  ent_coef_mode=("Auto", -6.0, 3e-4), device=0, seed=5)
a = B.Sac.build(cfg)
for _ in range(200): a.opt(rb)
a.sync()
h = []
for _ in range(20):   # short bursts from an idle queue: what the host pays per opt before any back-pressure
    a.sync()
    t0 = time.perf_counter()
    for _ in range(10): a.opt(rb)
    h.append((time.perf_counter() - t0) / 10 * 1e6)
    a.sync()
print(f"burst host_us_per_opt min={min(h):.1f} median={sorted(h)[10]:.1f}")
N = 2000
t0 = time.perf_counter()
for _ in range(N): a.opt(rb)
t1 = time.perf_counter()
a.sync()
t2 = time.perf_counter()
print(f"mode={'eager' if os.environ.get('BDR_NO_STEP_GRAPH') == '1' else 'graph'} host_enqueue_us={(t1 - t0) / N * 1e6:.1f} total_us={(t2 - t0) / N * 1e6:.1f}")
