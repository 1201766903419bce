#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sac.py tests/test_gpu_dqn.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/tests_gpu_b.log
echo "rc=${PIPESTATUS[0]}" >> gpurun_out/tests_gpu_b.log
tail -30 gpurun_out/tests_gpu_b.log
