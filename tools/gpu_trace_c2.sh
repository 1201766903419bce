#!/bin/bash
# kernel trace of the C2 step (serial schedule and default), for each environment variant given: gpurun_out/trace_<n>_{serial,default}.md
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
n=0
for v in "$@"; do
  n=$((n+1))
  for mode in serial default; do
    e="$v"; [ $mode = serial ] && e="$v BDR_SCHED=0"
    ( cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_t$n -o t -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_t$n.log 2>&1 )
    db=$(find $O/prof_t$n -name "*.db" | head -1)
    [ -n "$db" ] && { echo "# variant [$v] $mode"; python tools/rocprof_summary.py $db --skip-first 25; } > $O/trace_${n}_$mode.md 2>> $O/prof_t$n.log
    rm -rf $O/prof_t$n
  done
done
head -30 $O/trace_1_serial.md
