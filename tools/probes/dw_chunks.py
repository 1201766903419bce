"""A/B of the conv weight-gradient partial counts (BDR_DW_CHUNKS="c1,c2,c3"): per-kernel times and the whole step, same box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(2):
    for ch in sys.argv[1:] or ("", "256,32,32", "256,32,24", "256,16,16", "128,64,56", "128,32,24", "64,32,24"):
        env = dict(os.environ)
        if ch:
            env["BDR_DW_CHUNKS"] = ch
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2000", "--warmup", "200", "--no-cpu-baseline", "--profile-steps", "60",
                            "--capacity", "100000"], capture_output=True, text=True, env=env, cwd=ROOT)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            k = d["roofline"]["kernels_ms"]
            print(f"chunks [{ch:>10}]: step {d['ms_per_step'] * 1000:.1f} us ({d['value']:.0f}/s)  c1_dw {1000 * k['bwd_conv1_dw']:.1f} c2_dw {1000 * k['bwd_conv2_dw']:.1f} "
                  f"c3_dw {1000 * k['bwd_conv3_dw']:.1f} l1_dw {1000 * k['bwd_l1_dw']:.1f} reduce_adam {1000 * k['reduce_adam']:.1f}", flush=True)
        except Exception as e:  # noqa: BLE001
            print("chunks", ch, "failed", e, r.stderr[-300:])
