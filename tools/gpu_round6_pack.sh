#!/bin/bash
# round-6 evidence pack: bench lines (4 configs, with CPU baselines), rocprofv3 kernel traces (+ machine-readable twins), PMC passes
# (FETCH_SIZE, WRITE_SIZE in separate runs, --kernel-trace only) -> hbm_traffic.json stamped with the kernel sources' hash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
tag=${1:-r06}
for c in c2 c4 c5 c1; do
  timeout 900 python bench.py --config $c > $O/bench_${tag}_$c.json 2> $O/bench_${tag}_$c.err
done
timeout 600 python bench.py --config c2 --frame-ring --no-cpu-baseline > $O/bench_${tag}_c2_frame_ring.json 2>/dev/null
timeout 600 python bench.py --config c2 --per --no-cpu-baseline > $O/bench_${tag}_c2_per.json 2>/dev/null
BDR_NO_SAC_FUSE=1 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_layer_by_layer.json 2>/dev/null
BDR_SAC_SIDE_QUEUE=0 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_one_queue.json 2>/dev/null
BDR_NO_SAC_CHAIN=1 BDR_SAC_TAIL_IN_KERNEL=1 BDR_SAC_WAIT_PACKET=1 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_round4_form.json 2>/dev/null
timeout 600 python bench.py --config c4 --arithmetic f32_exact --no-cpu-baseline > $O/bench_${tag}_c4_exact_f32.json 2>/dev/null
timeout 600 python bench.py --config c2 --arithmetic f32_exact --no-cpu-baseline > $O/bench_${tag}_c2_exact_f32.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_c2_driver_form.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_c2_driver_form_2.json 2>/dev/null
timeout 600 python tools/probes/online_loop_device.py > $O/online_loop_device_${tag}.txt 2>/dev/null
trace() {  # name, env, bench args, json name (or -)
  ( cd /tmp && env $2 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_${tag}_$1 -o t -- python $GRAFT_REPO_ROOT/bench.py $3 --no-cpu-baseline --no-exact-leg --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_${tag}_$1.log 2>&1 )
  db=$(find $O/prof_${tag}_$1 -name "*.db" | head -1)
  if [ -n "$db" ]; then
    if [ "$4" != "-" ]; then python tools/rocprof_summary.py $db --skip-first 25 --json $O/$4 > $O/rocprof_${tag}_kernel_trace_$1.md 2>> $O/prof_${tag}_$1.log
    else python tools/rocprof_summary.py $db --skip-first 25 > $O/rocprof_${tag}_kernel_trace_$1.md 2>> $O/prof_${tag}_$1.log; fi
  fi
  rm -rf $O/prof_${tag}_$1
}
trace c2_serial "BDR_SCHED=0" "--config c2 --steps 200 --warmup 20" kernel_trace_c2_serial.json
trace c2_default "X=1" "--config c2 --steps 200 --warmup 20" -
trace c4 "X=1" "--config c4 --steps 20 --warmup 3" kernel_trace_c4_serial.json
trace c5 "X=1" "--config c5 --steps 200 --warmup 20" kernel_trace_c5_serial.json
trace c5_layer_by_layer "BDR_NO_SAC_FUSE=1" "--config c5 --steps 200 --warmup 20" -
trace c1 "X=1" "--config c1 --steps 500 --warmup 50" kernel_trace_c1_serial.json
pmc() {  # counter, pass name
  ( cd /tmp && BDR_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc $1 -d $GRAFT_REPO_ROOT/$O/pmc_${tag}_$2 -o $2 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-leg --profile-steps 1 > $GRAFT_REPO_ROOT/$O/pmc_${tag}_$2.log 2>&1 )
  db=$(find $O/pmc_${tag}_$2 -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db --skip-first 6 | sed -n '/PMC/,$p' > $O/pmc_${tag}_$2.md
  rm -rf $O/pmc_${tag}_$2
}
pmc FETCH_SIZE fetch
pmc WRITE_SIZE write
{ echo "# rocprofv3 --pmc passes of round 6 (B=256 Atari DQN step, serial schedule BDR_NO_OVERLAP=1; separate passes, --kernel-trace only)"; echo; echo "## FETCH_SIZE (KiB per dispatch)"; echo; sed -n '3,$p' $O/pmc_${tag}_fetch.md; echo; echo "## WRITE_SIZE (KiB per dispatch)"; echo; sed -n '3,$p' $O/pmc_${tag}_write.md; } > $O/rocprof_${tag}_pmc.md
python tools/make_hbm_traffic.py $O/pmc_${tag}_fetch.md $O/pmc_${tag}_write.md profiles/rocprof_${tag}_pmc.md > $O/hbm_traffic.json 2> $O/hbm_traffic.err
bash tools/gpu_round4_mfma.sh $tag > $O/mfma_pass.log 2>&1
ls $O | grep -E "$tag|hbm|kernel_trace" | head -60
# round 5 additions: what the longest k_gate of the default-schedule trace waited for (tools/gate_max.py reads the trace before it is deleted),
# the conv1 probes (forward forms, weight-gradient tilings, phase stamps)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_${tag}_gate -o t -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline --no-exact-leg --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_${tag}_gate.log 2>&1 )
db=$(find $O/prof_${tag}_gate -name "*.db" | head -1)
[ -n "$db" ] && python tools/gate_max.py $db > $O/gate_max_${tag}.md 2>&1
rm -rf $O/prof_${tag}_gate
( cd tools/probes && { ./conv1_rw_probe.bin 256; ./c1dw_probe.bin 256 4; ./c1dw_trace.bin; ./mfma_bf16_rate.bin; } > ../../$O/conv1_probes_${tag}.txt 2>&1 )
( cd tools/probes && timeout 120 ./chain2_probe.bin > ../../$O/chain2_probe_${tag}.txt 2>&1 )
timeout 300 python tools/probes/sample_device_probe.py 1000 > $O/sample_device_${tag}.txt 2>/dev/null
timeout 300 python tools/probes/sample_latency.py > $O/sample_latency_${tag}.txt 2>/dev/null
ls $O | grep -E "$tag" | head -80
