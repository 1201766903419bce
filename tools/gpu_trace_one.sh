#!/bin/bash
# one rocprofv3 kernel trace of bench.py: gpu_trace_one.sh <name> "<bench args>" [ENV=VAL]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
name=$1; args=$2; envs=${3:-X=1}
( cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $args --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1 )
db=$(find $O/prof_$name -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db --skip-first 25 > $O/rocprof_kernel_trace_$name.md 2>> $O/prof_$name.log
find $O/prof_$name -name "*.db" -delete
cat $O/rocprof_kernel_trace_$name.md | head -40
