#!/bin/bash
# first GPU call of round 2: tests, the four bench configs, rocprofv3 kernel traces for c2 / c4 / c5 / c1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/r2a_tests.log
echo "tests rc=$?" >> $O/r2a_tests.log
for c in c2 c4 c5 c1; do
  timeout 600 python bench.py --config $c > $O/r2a_bench_$c.json 2> $O/r2a_bench_$c.err
done
for c in c2 c4 c5 c1; do
  case $c in c2) S="--steps 200 --warmup 20";; c4) S="--steps 20 --warmup 3";; c5) S="--steps 200 --warmup 20";; c1) S="--steps 500 --warmup 50";; esac
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r02_$c -o r02_$c -- python $GRAFT_REPO_ROOT/bench.py --config $c $S --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/r2a_prof_$c.log 2>&1 )
  db=$(find $O/prof_r02_$c -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db --skip-first 25 > $O/rocprof_r02_kernel_trace_$c.md 2>> $O/r2a_prof_$c.log
  # keep the merge-back small: drop the raw databases
  find $O/prof_r02_$c -name "*.db" -size +20M -delete
done
ls -la $O | tail -30
