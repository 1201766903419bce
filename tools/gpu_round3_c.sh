#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sac.py tests/test_gpu_async_trainer.py tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/tests_gpu_c.log
echo "rc=${PIPESTATUS[0]}" >> gpurun_out/tests_gpu_c.log
tail -25 gpurun_out/tests_gpu_c.log
for v in "X=1" "BDR_SAC_SIDE_QUEUE=0" "X=1" "BDR_SAC_SIDE_QUEUE=0"; do
  echo "== [$v]"
  env $v timeout 300 python bench.py --config c5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
