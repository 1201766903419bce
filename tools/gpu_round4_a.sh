#!/bin/bash
# round 4, call A: conv2-dX shape sweep (probe), DQN tests on the new default library, whole-step A/B of library variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
[ -n "$SKIP_SWEEP" ] || timeout 300 tools/probes/dx2_sweep.bin > $O/r4_dx2_sweep.log 2>&1
[ -n "$SKIP_TESTS" ] || timeout 1200 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_per.py -m gpu -q -x 2>&1 | tail -15 > $O/r4a_tests.log
tail -3 $O/r4a_tests.log
: > $O/r4a_ab.log
V=$GRAFT_REPO_ROOT/scratch/variants
for rep in 1 2; do
  for v in "" "$@"; do
    lib=""; [ -n "$v" ] && lib="BORDER_AMD_LIB=$V/lib_$v.so"
    [ $rep = 1 ] && { echo -n "[$v] " >> $O/r4a_ab.log; env $lib X=1 timeout 120 python tools/probes/variant_check.py 2>&1 | tail -1 >> $O/r4a_ab.log; }
    echo -n "[$v] rep $rep: " >> $O/r4a_ab.log
    env $lib X=1 timeout 300 python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/r4a_ab.log 2>&1
  done
done
cat $O/r4a_ab.log
[ -n "$SKIP_SWEEP" ] || cat $O/r4_dx2_sweep.log
