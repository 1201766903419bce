#!/bin/bash
# Round-5 same-box A/B: optional test files (TESTS="..."), then bench c2 steady state for every environment variant given as an argument
# ("" = default; PREV = the round-4 library scratch/ab/libborder_amd_prev.so built by tools/probes/build_prev.sh), two interleaved repetitions.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then
  timeout 1500 python -m pytest $TESTS -m gpu -q -x 2>&1 | tail -30 > gpurun_out/ab_tests.log
  tail -3 gpurun_out/ab_tests.log
fi
: > gpurun_out/ab_bench.log
for rep in 1 2; do
  for v in "$@"; do
    e="$v"; [ "$v" = "PREV" ] && e="BORDER_AMD_LIB=$PWD/scratch/ab/libborder_amd_prev.so"
    echo "== variant: [$v] rep $rep" >> gpurun_out/ab_bench.log
    env $e python bench.py --config ${CONFIG:-c2} --steps ${STEPS:-3000} --warmup 200 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/ab_bench.log 2>&1
  done
done
cat gpurun_out/ab_bench.log
