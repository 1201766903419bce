#!/bin/bash
# same-box A/B of library variants (scratch/variants/lib_*.so) on bench c2; two passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/variants.log
for rep in 1 2; do
  for lib in border_amd/libborder_amd.so "$@"; do
    r=$(BORDER_AMD_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$rep $lib $r" >> gpurun_out/variants.log
  done
done
cat gpurun_out/variants.log
