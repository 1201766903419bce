#!/bin/bash
# usage: sweep_env.sh VAR "v1 v2 ..." kernel_key  -> opt-steps/s and that kernel's time per setting (same box)
var=$1; vals=$2; key=$3
for v in $vals; do
  env $var=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/sweep.json
  python - "$var" "$v" "$key" <<'PY'
import json, sys
d = json.load(open('/tmp/sweep.json')); k = d["roofline"]["kernels_ms"]
print(sys.argv[1], sys.argv[2], d["value"], sys.argv[3], k.get(sys.argv[3]), "reduce", k.get("bwd_conv_reduce"))
PY
done
