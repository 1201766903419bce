cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d gpurun_out/prof_tlper -o tlper -- python bench.py --per --steps 300 --warmup 20 --no-cpu-baseline --profile-steps 1 > gpurun_out/tlper.log 2>&1
python tools/rocprof_timeline.py gpurun_out/prof_tlper/tlper_results.db --skip 60 --first-kernel k_per_sample > gpurun_out/timeline_per.md 2>&1
rm -rf gpurun_out/prof_tlper
cat gpurun_out/timeline_per.md
