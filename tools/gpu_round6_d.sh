#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6d
python tools/diag/amsgrad_cnn.py > gpurun_out/r6d/amsgrad_cnn.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_multi.py -m gpu -q -x > gpurun_out/r6d/pytest_sac.log 2>&1; echo "rc $?" >> gpurun_out/r6d/pytest_sac.log
for i in 1 2; do python bench.py --config c5 --no-cpu-baseline > gpurun_out/r6d/bench_c5_$i.json 2>/dev/null; done
cat gpurun_out/r6d/amsgrad_cnn.txt; tail -3 gpurun_out/r6d/pytest_sac.log
