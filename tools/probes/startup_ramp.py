"""How long does the C2 step take right after start-up?  Windows of 5 opt steps (device-synchronised) from a cold agent."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import argparse
import border_amd as B
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--order", default="fill_first"); ap.add_argument("--idle-ms", type=float, default=0)
a = ap.parse_args()
class A: capacity = int(os.environ.get("CAP", "200000")); batch = None; per = False; frame_ring = False; double_dqn = False; loss = "SmoothL1"
conf = bench.build_config(B, "c2", A, 0, 0)
agent, rb = conf["agent"], conf["rb"]
agent.train()
if a.order == "refill":   # GPU busy (ring fill: ~11 GB of writes) right before the first step
    agent.sync()
    rb.fill_synthetic(A.capacity, seed=0, kind=0, n_actions=6)
if a.idle_ms: time.sleep(a.idle_ms / 1e3)
out = []
for w in range(16):
    t0 = time.perf_counter()
    for _ in range(5): agent.opt(rb)
    agent.sync()
    out.append((time.perf_counter() - t0) / 5 * 1e3)
print("ms/step per window of 5:", " ".join(f"{x:.3f}" for x in out))
t0 = time.perf_counter()
for _ in range(400): agent.opt(rb)
agent.sync()
print(f"steady: {(time.perf_counter() - t0) / 400 * 1e3:.4f} ms/step")
for idle in (1, 10, 100):
    time.sleep(idle / 1e3)
    t0 = time.perf_counter()
    for _ in range(20): agent.opt(rb)
    agent.sync()
    print(f"after {idle} ms idle: 20 steps at {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms/step")
time.sleep(0.01)
hs = []
t00 = time.perf_counter()
for _ in range(25):
    t0 = time.perf_counter(); agent.opt(rb); hs.append((time.perf_counter() - t0) * 1e6)
t_enq = time.perf_counter() - t00
agent.sync()
t_all = time.perf_counter() - t00
print("host us per opt call (25 after 10 ms idle):", " ".join(f"{x:.0f}" for x in hs))
print(f"enqueue done after {t_enq * 1e3:.3f} ms, GPU done after {t_all * 1e3:.3f} ms -> {t_all / 25 * 1e3:.4f} ms/step")
