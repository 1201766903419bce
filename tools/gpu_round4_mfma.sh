#!/bin/bash
# MFMA-busy counter pass (north_star: "rocprof ... MFMA-busy counters against peak"): one --pmc pass over the C2 step (serial schedule),
# one over C4, one over the calibration kernels; --kernel-trace only.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
tag=${1:-r04}
CNT="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"
( cd /tmp && BDR_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_${tag}_mfma -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-leg --profile-steps 1 > $GRAFT_REPO_ROOT/$O/pmc_${tag}_mfma.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_${tag}_mfma_c4 -o m -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-exact-leg --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_${tag}_mfma_c4.log 2>&1 )
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_${tag}_mfma_cal -o m -- $GRAFT_REPO_ROOT/tools/probes/mfma_busy_cal.bin > $GRAFT_REPO_ROOT/$O/pmc_${tag}_mfma_cal.log 2>&1 )
db=$(find $O/pmc_${tag}_mfma -name "*.db" | head -1); db4=$(find $O/pmc_${tag}_mfma_c4 -name "*.db" | head -1); dbc=$(find $O/pmc_${tag}_mfma_cal -name "*.db" | head -1)
{ python tools/mfma_busy.py $db $dbc; echo; echo "## C4 (IQN, batch 512, 64 quantiles)"; echo; python tools/mfma_busy.py $db4 $dbc | sed -n '5,$p'; } > $O/rocprof_${tag}_mfma_busy.md 2> $O/mfma_busy.err
cat $O/rocprof_${tag}_mfma_busy.md; tail -3 $O/mfma_busy.err $O/pmc_${tag}_mfma_cal.log
rm -rf $O/pmc_${tag}_mfma $O/pmc_${tag}_mfma_c4 $O/pmc_${tag}_mfma_cal
