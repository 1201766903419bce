"""Policy::sample on device-resident rows, n_procs = 1 / 4 / 16: wall time per call (and, under rocprofv3 --kernel-trace, its kernels)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import border_amd as B

ROW = 4 * 84 * 84
rng = np.random.default_rng(0)
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  soft_update_interval=10000, batch_size=256, critic_loss="SmoothL1", device=0, param_seed=0, train=True)
a = B.Dqn.build(cfg)
a.set_explorer(B.EpsilonGreedy.with_final_step(1_000_000), seed=1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for n in (1, 4, 16):
    prep = B.AtariPreprocessor(n)
    ixs = np.arange(n)
    pool = [rng.integers(0, 256, (n, 210, 160, 3), dtype=np.uint8) for _ in range(3)]
    prep.reset_device(ixs, pool[0]); prep.step_device(ixs, pool[1], pool[2])
    st = prep.device_stacks()
    for _ in range(50): a.sample_device(st, n, ROW)
    t0 = time.perf_counter()
    for _ in range(reps): a.sample_device(st, n, ROW)
    dt = (time.perf_counter() - t0) / reps * 1e6
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100_000, seed=42), (4, 1, 84, 84), "uint8")
    act = np.zeros((n, 1), np.int64); rew = np.zeros(n, np.float32); fl = np.zeros(n, np.int8)
    for _ in range(20): rb.push_device(prep.device_prev_stacks(), ROW, act, st, ROW, rew, fl, fl)
    t0 = time.perf_counter()
    for _ in range(reps // 4): rb.push_device(prep.device_prev_stacks(), ROW, act, st, ROW, rew, fl, fl)
    dp = (time.perf_counter() - t0) / (reps // 4) * 1e6
    print(f"n_procs={n:3d}: bdr_agent_sample_device {dt:7.1f} us   bdr_replay_push_device {dp:7.1f} us", flush=True)
    rb.close(); prep.close()
a.close()
