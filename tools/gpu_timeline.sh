#!/bin/bash
# kernel-trace timeline of the default schedule (c2): tools/rocprof_timeline.py on a rocprofv3 trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
tag=${1:-split}
( cd /tmp && env ${@:2} timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_tl_$tag -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-exact-leg --profile-steps 0 > $GRAFT_REPO_ROOT/$O/tl_$tag.log 2>&1 )
db=$(find $O/prof_tl_$tag -name "*.db" | head -1)
python tools/rocprof_timeline.py $db --skip 40 > $O/timeline_$tag.md 2>> $O/tl_$tag.log
find $O/prof_tl_$tag -name "*.db" -size +20M -delete
tail -3 $O/tl_$tag.log | cut -c1-300
cat $O/timeline_$tag.md
