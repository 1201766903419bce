#!/bin/bash
# A/B on one box: DQN + PER tests, then bench c2 with the given environment variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_per.py tests/test_gpu_trainer.py tests/test_gpu_sync_dp.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/ab_tests.log
tail -3 gpurun_out/ab_tests.log
: > gpurun_out/ab_bench.log
for rep in 1 2; do
  for v in "" "BDR_NO_SPLIT_FWD=1" "$@"; do
    echo "== variant: [$v] rep $rep" >> gpurun_out/ab_bench.log
    env $v python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/ab_bench.log 2>&1
  done
done
cat gpurun_out/ab_bench.log
