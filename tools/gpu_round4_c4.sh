#!/bin/bash
# round 4, C4: IQN tests, then bench c4 with the split-operand merge layer vs the exact FP32-MFMA kernels (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_iqn.py -m gpu -q -x 2>&1 | tail -15 > $O/r4c4_tests.log
tail -4 $O/r4c4_tests.log
for rep in 1 2; do
  for v in "X=1" "BDR_IQN_F32_EXACT=1"; do
    echo -n "[$v] rep $rep: "
    env $v timeout 600 python bench.py --config c4 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print(d['value'], d['ms_per_step'], d['dtype'][:20], {x: round(k[x],4) for x in k if 'f_fwd1' in x or 'f_dx1' in x or 'f_dw1' in x})"
  done
done 2>&1 | tee $O/r4c4_ab.log
