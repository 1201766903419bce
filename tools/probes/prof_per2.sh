cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_per -o per -- python bench.py --steps 300 --warmup 20 --no-cpu-baseline --profile-steps 0 --per > gpurun_out/benchprof_per.log 2>&1
db=$(find gpurun_out/prof_per -name "*.db" | head -1)
python tools/rocprof_summary.py $db --skip-first 21 | cut -c1-120 | head -40
rm -rf gpurun_out/prof_per
