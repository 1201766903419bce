#!/bin/bash
# round 6: conv2 dX position-class tiles - probe (isolation, bit identity), then the suite and the driver-form line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6b
timeout 300 tools/probes/dx2_pos_probe.bin > gpurun_out/r6b/dx2_pos_probe.txt 2>&1
python -m pytest tests -m gpu -q -x > gpurun_out/r6b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6b/pytest.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/bench_c2_$i.json 2> gpurun_out/r6b/bench_c2_$i.err; done
cat gpurun_out/r6b/dx2_pos_probe.txt; tail -3 gpurun_out/r6b/pytest.log
