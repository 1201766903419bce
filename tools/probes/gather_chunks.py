"""k_gather time (C2 shape: 256 samples x 2 x 28 224 B out of a 1 M-transition ring) for 1..8 workgroups per sample
(BDR_GATHER_CHUNKS, read once per process), from bench.py's per-kernel HIP-event brackets."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for ch in ("0", "1", "2", "3", "4", "6", "8"):
    env = dict(os.environ, BDR_GATHER_CHUNKS=ch)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "50", "--no-cpu-baseline", "--profile-steps", "60"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        k = d["roofline"]["kernels_ms"]
        print(f"chunks {ch}: sample {1000 * k['sample']:.2f} us, fwd_conv1 {1000 * k['fwd_conv1']:.2f} us, step {d['ms_per_step'] * 1000:.1f} us")
    except Exception as e:  # noqa: BLE001
        print("chunks", ch, "failed", e, r.stderr[-300:])
