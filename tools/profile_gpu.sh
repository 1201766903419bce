#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace profile of the bench + summary table.
#   gpurun -- 'bash tools/profile_gpu.sh r01b'
# Per-kernel durations are taken with the dW/dX stream overlap OFF (BDR_NO_OVERLAP=1) so that
# concurrently running kernels do not inflate each other's durations; the bench line printed first
# is the normal (overlapped) run.
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/bench_$tag.log 2>/dev/null
grep -a metric gpurun_out/bench_$tag.log | cut -c1-140
BDR_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 1 > gpurun_out/benchprof_$tag.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_$tag/${tag}_results.db --skip-first 21 > gpurun_out/rocprof_$tag.md
cat gpurun_out/rocprof_$tag.md
rm -rf gpurun_out/prof_$tag
