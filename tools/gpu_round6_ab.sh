#!/bin/bash
# Round-6 same-box A/B: bench c2 over --steps 3000 (plain protocol, no legs) for every variant given as an argument, REPS interleaved repetitions.
#   ""            the in-tree library with the default environment
#   LIB:<name>    scratch/ab/libborder_amd_<name>.so (tools/probes/build_dqn_variant.sh)
#   K=V[+K=V]     environment variables for the in-tree library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/${OUTNAME:-ab_r6}.log
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then
  timeout 1500 python -m pytest $TESTS -m gpu -q -x 2>&1 | tail -30 > gpurun_out/ab_tests.log
  tail -3 gpurun_out/ab_tests.log
fi
: > $OUT
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    e=$(echo "$v" | tr '+' ' ')
    case "$v" in LIB:*) e="BORDER_AMD_LIB=$PWD/scratch/ab/libborder_amd_${v#LIB:}.so";; esac
    echo -n "[$v] rep $rep: " >> $OUT
    env $e python bench.py --config ${CONFIG:-c2} --steps ${STEPS:-3000} --warmup 200 --no-cpu-baseline --profile-steps 0 --no-exact-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $OUT 2>&1
  done
done
cat $OUT
