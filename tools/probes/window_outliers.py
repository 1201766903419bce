"""The driver's 5 + 20-step window, many times in one process: distribution of the window rate, and the longest single Agent::opt call
(host side) inside every window - is an outlier window a host stall?"""
import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import bench, border_amd as B

class A: pass
args = A(); args.loss = "SmoothL1"; args.per = False; args.frame_ring = False; args.capacity = None; args.batch = None; args.double_dqn = False
for k, v in dict(seed=0, n_actions=6).items(): setattr(args, k, v)
import argparse
ns = argparse.Namespace(config="c2", loss="SmoothL1", per=False, frame_ring=False, capacity=None, batch=None, double_dqn=False, overlap_exchange=False, sync_interval=1)
try:
    conf = bench.build_config(B, "c2", ns, 0, 0)
except Exception as e:
    print("build_config signature differs:", e); raise
agent, rb = conf["agent"], conf["rb"]
agent.train()
for _ in range(300): agent.opt(rb)
agent.sync()
gcoff = len(sys.argv) > 1 and sys.argv[1] == "nogc"
if gcoff: gc.disable()
rates, worst = [], []
for w in range(60):
    for _ in range(5): agent.opt(rb)
    agent.sync()
    t0 = time.perf_counter(); mx = 0.0
    for _ in range(20):
        a = time.perf_counter(); agent.opt(rb); b = time.perf_counter(); mx = max(mx, b - a)
    agent.sync()
    dt = time.perf_counter() - t0
    rates.append(20 / dt); worst.append(mx * 1e6)
    time.sleep(0.002 * (w % 3))
r = np.array(rates); wv = np.array(worst)
print(f"gc {'off' if gcoff else 'on'}: window rate min {r.min():.0f} p10 {np.percentile(r, 10):.0f} median {np.median(r):.0f} max {r.max():.0f}; longest opt() call per window: median {np.median(wv):.0f} us, max {wv.max():.0f} us")
for i in np.argsort(r)[:5]: print(f"   window {i}: {r[i]:.0f} opt-steps/s, longest opt() call {wv[i]:.0f} us")
agent.close(); rb.close()
