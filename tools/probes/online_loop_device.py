"""Actor-side cost of one vectorised env step, host-pointer path vs device-resident path (round 4):
   Policy::sample for n_procs observations, the push of n transitions, and the whole loop
   [sample -> emulator frames -> bdr_atari_prep step -> push (-> opt every step)] for a 256-env synthetic vectorised env.
The emulator is a pool of pre-rendered RGB frames (its cost is not what is measured); the raw frames cross PCIe in both paths."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import border_amd as B

ROW = 4 * 84 * 84
rng = np.random.default_rng(0)


def timeit(f, n=30, warm=5):
    for _ in range(warm): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return 1e6 * (time.perf_counter() - t0) / n


cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  soft_update_interval=10000, batch_size=256, critic_loss="SmoothL1", device=0, param_seed=0, train=True)
a = B.Dqn.build(cfg)
a.set_explorer(B.EpsilonGreedy.with_final_step(1_000_000), seed=1)
for n in (1, 256):
    prep = B.AtariPreprocessor(n)
    ixs = np.arange(n)
    pool = [rng.integers(0, 256, (n, 210, 160, 3), dtype=np.uint8) for _ in range(3)]
    prep.reset_device(ixs, pool[0])
    prep.step_device(ixs, pool[1], pool[2])
    obs = prep.obs(ixs)
    t_h = timeit(lambda: a.sample(obs))
    t_d = timeit(lambda: a.sample_device(prep.device_stacks(), n, ROW))
    t_ro = timeit(lambda: prep.obs(ixs))
    print(f"n_procs={n:4d}: bdr_agent_sample {t_h:8.1f} us (host rows)   bdr_agent_sample_device {t_d:8.1f} us   (reading the stacks back for the host path: {t_ro:8.1f} us)")
    rb_h = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100_000, seed=42), (4, 1, 84, 84), "uint8")
    rb_d = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100_000, seed=42), (4, 1, 84, 84), "uint8")
    act = np.zeros((n, 1), np.int64); rew = np.zeros(n, np.float32); fl = np.zeros(n, np.int8)
    t_ph = timeit(lambda: rb_h.push(obs, act, obs, rew, fl, fl))
    t_pd = timeit(lambda: rb_d.push_device(prep.device_prev_stacks(), ROW, act, prep.device_stacks(), ROW, rew, fl, fl))
    print(f"            push of {n} transitions: bdr_replay_push {t_ph:8.1f} us   bdr_replay_push_device {t_pd:8.1f} us")
    # the loop, with an opt step per iteration (B = 256) once the buffer holds 2 000 transitions
    for path in ("host", "device"):
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100_000, seed=42), (4, 1, 84, 84), "uint8")
        rb.fill_synthetic(2000, seed=0, kind=0, n_actions=6)
        prep.reset_device(ixs, pool[0])
        iters = 60 if n == 256 else 300
        cur = prep.obs(ixs) if path == "host" else None
        def it(k):
            global cur
            if path == "host":
                act = a.sample(cur).reshape(n, 1)
                nobs = prep.step(ixs, pool[k % 3], pool[(k + 1) % 3])
                rb.push(cur, act, nobs, rew, fl, fl)
                cur = nobs
            else:
                act = a.sample_device(prep.device_stacks(), n, ROW).reshape(n, 1)
                prep.step_device(ixs, pool[k % 3], pool[(k + 1) % 3])
                rb.push_device(prep.device_prev_stacks(), ROW, act, prep.device_stacks(), ROW, rew, fl, fl)
            a.opt(rb)
        for k in range(5): it(k)
        a.sync()
        t0 = time.perf_counter()
        for k in range(iters): it(k)
        a.sync()
        dt = time.perf_counter() - t0
        print(f"            loop ({path:6s} rows): {iters / dt:8.1f} iterations/s = {n * iters / dt:10.1f} env steps/s with one opt (B=256) per iteration")
        rb.close()
    rb_h.close(); rb_d.close(); prep.close()
a.close()
