import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import border_amd as B
rng = np.random.default_rng(0)
a = B.Iqn.build(B.IqnConfig(n_actions=6, device=0, batch_size=32))
a.eval()
for n in (1, 4):
    obs = rng.integers(0, 255, (n, 4, 1, 84, 84)).astype(np.uint8)
    for _ in range(50): a.sample(obs)
    t0 = time.perf_counter()
    for _ in range(500): a.sample(obs)
    print(f"iqn cnn n={n}: sample {(time.perf_counter() - t0) / 500 * 1e6:.1f} us")
a.close()
