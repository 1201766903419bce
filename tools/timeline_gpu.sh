#!/bin/bash
# Runs ON the GPU box: kernel trace of the overlapped (two-stream) opt step -> per-step timeline with queue gaps.
#   gpurun -- 'bash tools/timeline_gpu.sh tag'
tag=${1:-tl}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 300 --warmup 20 --no-cpu-baseline --profile-steps 1 > gpurun_out/$tag.log 2>&1
grep -a metric gpurun_out/$tag.log | cut -c1-140
python tools/rocprof_timeline.py gpurun_out/prof_$tag/${tag}_results.db --skip 60 > gpurun_out/timeline_$tag.md 2>&1
rm -rf gpurun_out/prof_$tag
cat gpurun_out/timeline_$tag.md
