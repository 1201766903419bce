#!/bin/bash
# round 4: whole-step A/B of environment variants on one box: fingerprint of a seeded opt stream + 2 x bench c2 (3000 steps) per variant
# usage: tools/gpu_round4_env.sh "VAR=1" "VAR=2 OTHER=x" ...   (the empty variant always runs first)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
: > $O/r4_env_ab.log
for rep in 1 2; do
  for v in "X=1" "$@"; do
    [ $rep = 1 ] && { echo -n "[$v] " >> $O/r4_env_ab.log; env $v timeout 120 python tools/probes/variant_check.py 2>&1 | tail -1 >> $O/r4_env_ab.log; }
    echo -n "[$v] rep $rep: " >> $O/r4_env_ab.log
    env $v timeout 300 python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/r4_env_ab.log 2>&1
  done
done
cat $O/r4_env_ab.log
