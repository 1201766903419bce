#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace profile of the bench + summary table.
#   gpurun -- 'bash tools/profile_gpu.sh r01b'
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 1 > gpurun_out/bench_$tag.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_$tag/${tag}_results.db --skip-first 21 > gpurun_out/rocprof_$tag.md
tail -1 gpurun_out/bench_$tag.log | cut -c1-300
cat gpurun_out/rocprof_$tag.md
rm -rf gpurun_out/prof_$tag
