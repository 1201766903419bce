"""Soak of the SAC two-queue sequence at the config-5 shape: N opts over a ring that keeps growing, two queues (two-layer launches, in-kernel prologue wait,
batch-wide sums in the next launch) against ONE queue - every parameter, moment and the buffer's next indices bit for bit.  A stale read anywhere in the
flag-ordered hand-overs shows up as a difference.  usage: python tools/probes/sac_soak.py [n_opts] [batch]"""
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import border_amd as B
n, bs, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
od, ad = 17, 6
rng = np.random.default_rng(5)
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=20000, seed=9), (od,), np.float32, (ad,), np.float32)
def push(k):
    rb.push(rng.standard_normal((k, od)).astype(np.float32), rng.uniform(-1, 1, (k, ad)).astype(np.float32), rng.standard_normal((k, od)).astype(np.float32),
            rng.standard_normal(k).astype(np.float32), (rng.random(k) < .05).astype(np.int8), np.zeros(k, np.int8))
push(4000)
a = B.Sac.build(B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(256, 256), q_units=(256, 256), n_critics=2, batch_size=bs, ent_coef_mode=("Auto", -6.0, 3e-4),
                            n_updates_per_opt=1, device=0, seed=3))
a.train()
for k in range(n):
    a.opt(rb)
    if k %% 64 == 0: push(50)
names = ["pi", "log_alpha", "qnet_0", "qnet_1", "qnet_tgt_0", "qnet_tgt_1"]
res = {m: a.get_params(m) for m in names}
res["pi_v"] = a.get_params("pi", "exp_avg_sq"); res["next"] = rb.sample_indices(64)
np.savez(out, **res)
print("finite", bool(np.isfinite(res["pi"]).all()), "n_opts", a.n_opts)
a.close(); rb.close()
"""

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    outs = {}
    for name, env in (("two_queues", {}), ("one_queue", {"BDR_SAC_SIDE_QUEUE": "0"}), ("two_queues_again", {})):
        e = dict(os.environ); e.update(env); e["BDR_NO_STEP_GRAPH"] = "1"
        path = f"/tmp/sac_soak_{name}.npz"
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(n), str(bs), path], env=e, capture_output=True, text=True, timeout=3000)
        print(name, r.stdout.strip(), r.stderr.strip()[-300:])
        outs[name] = np.load(path)
    ref = outs["one_queue"]
    bad = 0
    for name in ("two_queues", "two_queues_again"):
        for k in ref.files:
            same = bool((outs[name][k] == ref[k]).all())
            bad += not same
            if not same: print("DIFFERENT", name, k, np.abs(outs[name][k].astype(np.float64) - ref[k]).max())
    print(f"{n} opts at B = {bs}: {'ALL EQUAL' if bad == 0 else str(bad) + ' arrays differ'}")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
