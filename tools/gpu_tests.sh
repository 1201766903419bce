#!/bin/bash
# GPU test run: every -m gpu test (no -x), tail of the report into gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -150 > gpurun_out/tests_gpu.log
echo "rc=${PIPESTATUS[0]}" >> gpurun_out/tests_gpu.log
tail -5 gpurun_out/tests_gpu.log
