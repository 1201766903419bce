#!/bin/bash
# evidence pack: bench lines for the four configs (with CPU baselines), rocprofv3 kernel traces (c2 serial + default schedule, c4, c5, c1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
tag=${1:-r02}
for c in c2 c4 c5 c1; do
  timeout 900 python bench.py --config $c > $O/${tag}_bench_$c.json 2> $O/${tag}_bench_$c.err
done
timeout 600 python bench.py --config c2 --frame-ring --no-cpu-baseline > $O/${tag}_bench_c2_frame_ring.json 2>/dev/null
timeout 600 python bench.py --config c2 --per --no-cpu-baseline > $O/${tag}_bench_c2_per.json 2>/dev/null
trace() {  # name, env, bench args
  ( cd /tmp && env $2 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_${tag}_$1 -o t -- python $GRAFT_REPO_ROOT/bench.py $3 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/${tag}_prof_$1.log 2>&1 )
  db=$(find $O/prof_${tag}_$1 -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db --skip-first 25 > $O/rocprof_${tag}_kernel_trace_$1.md 2>> $O/${tag}_prof_$1.log
  find $O/prof_${tag}_$1 -name "*.db" -delete
}
trace c2_serial "BDR_SCHED=0" "--config c2 --steps 200 --warmup 20"
trace c2_default "X=1" "--config c2 --steps 200 --warmup 20"
trace c4 "X=1" "--config c4 --steps 20 --warmup 3"
trace c5 "X=1" "--config c5 --steps 200 --warmup 20"
trace c1 "X=1" "--config c1 --steps 500 --warmup 50"
ls $O | grep $tag | head -40
