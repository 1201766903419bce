#!/bin/bash
# round 6, first pass: the full GPU suite on the new boundary (arithmetic field, AdamW for IQN / SAC, nearest split), then the driver-form bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6a
python tools/diag/golden_b8_modes.py > gpurun_out/r6a/golden_b8_modes.txt 2>&1
python -m pytest tests -m gpu -q > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r6a/bench_c2_driver.json 2> gpurun_out/r6a/bench_c2_driver.err
tail -3 gpurun_out/r6a/pytest.log; cat gpurun_out/r6a/golden_b8_modes.txt
