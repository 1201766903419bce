#!/bin/bash
# Runs ON the GPU box: PMC counter passes (separate runs, kernel-trace only) for the bench kernels.
#   gpurun -- 'bash tools/pmc_gpu.sh r01 "SQ_WAVES SQ_BUSY_CYCLES ..." pass1'
tag=${1:-run}; counters=${2}; pass=${3:-p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc $counters -d gpurun_out/pmc_${tag}_$pass -o $pass -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 > gpurun_out/pmc_${tag}_$pass.log 2>&1
python tools/rocprof_summary.py gpurun_out/pmc_${tag}_$pass/${pass}_results.db --skip-first 6 > gpurun_out/pmc_${tag}_$pass.md
cat gpurun_out/pmc_${tag}_$pass.md | sed -n '/PMC/,$p'
rm -rf gpurun_out/pmc_${tag}_$pass
