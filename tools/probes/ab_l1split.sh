for rep in 1 2; do
for lib in "" _l1s8 _l1s14; do
  if [ -n "$lib" ]; then export BORDER_AMD_LIB=$PWD/border_amd/libborder_amd$lib.so; else unset BORDER_AMD_LIB; fi
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/sweep.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open('/tmp/sweep.json')); k = d["roofline"]["kernels_ms"]
print("lib", sys.argv[1] or "base(7)", d["value"], "fwd_l1", k.get("fwd_l1"), "head", k.get("head_fwd_td"))
PY
done; done
