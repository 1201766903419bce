#!/bin/bash
# usage: ab_libs.sh kernel_key lib1.so lib2.so ...   ("base" = the in-tree library); two interleaved repetitions on one box
key=$1; shift
for rep in 1 2; do
for lib in base "$@"; do
  if [ "$lib" != base ]; then export BORDER_AMD_LIB=$PWD/$lib; else unset BORDER_AMD_LIB; fi
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/sweep.json
  python - "$lib" "$key" <<'PY'
import json, sys
d = json.load(open('/tmp/sweep.json')); k = d["roofline"]["kernels_ms"]
print(sys.argv[1], d["value"], {x: k.get(x) for x in sys.argv[2].split(',')})
PY
done; done
