"""What a 20-step window spends beyond 20 steady steps: the cost of Agent sync on an idle agent, the enqueue time of the K steps, the
final sync, and the same window at K = 20 / 100 / 500 (a fixed per-window cost shows as a 1/K term)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import argparse
import bench, border_amd as B

ns = argparse.Namespace(config="c2", loss="SmoothL1", per=False, frame_ring=False, capacity=200000, batch=None, double_dqn=False, overlap_exchange=False, sync_interval=1)
conf = bench.build_config(B, "c2", ns, 0, 0)
agent, rb = conf["agent"], conf["rb"]
agent.train()
for _ in range(500): agent.opt(rb)
agent.sync()
t = []
for _ in range(50):
    a = time.perf_counter(); agent.sync(); t.append(time.perf_counter() - a)
print(f"sync of an idle agent: median {1e6 * np.median(t):.1f} us, min {1e6 * min(t):.1f}")
for K in (20, 100, 500, 2000):
    rows = []
    for w in range(30 if K <= 100 else 6):
        for _ in range(5): agent.opt(rb)
        agent.sync()
        t0 = time.perf_counter()
        for _ in range(K): agent.opt(rb)
        t1 = time.perf_counter()
        agent.sync()
        t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t2 - t0))
    r = np.array(rows)
    med = np.median(r, axis=0)
    print(f"K = {K:4d}: enqueue {1e6 * med[0]:8.1f} us, final sync {1e6 * med[1]:8.1f} us, window {1e6 * med[2]:9.1f} us = {1e6 * med[2] / K:6.2f} us/step = {K / med[2]:7.1f} opt-steps/s"
          f" (best {K / r[:, 2].min():7.1f}, worst {K / r[:, 2].max():7.1f})")
agent.close(); rb.close()
