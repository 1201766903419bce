cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BDR_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_per -o per -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 1 --per > gpurun_out/benchprof_per.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_per/per_results.db --skip-first 21 | grep -E "k_per|k_gather|kernel" 
rm -rf gpurun_out/prof_per
