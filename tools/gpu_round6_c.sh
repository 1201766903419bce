#!/bin/bash
# round 6: full suite on the new conv2 dX default + the two-rank tests over the communicator's host-transport build, the 2-ranks-on-one-GPU bench flow
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6c/pytest.log
BDR_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 40 --warmup 5 --capacity 20000 > gpurun_out/r6c/bench_share2.json 2> gpurun_out/r6c/bench_share2.err
BDR_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 40 --warmup 5 --capacity 20000 --overlap-exchange > gpurun_out/r6c/bench_share2_overlap.json 2> gpurun_out/r6c/bench_share2_overlap.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r6c/bench_c2_driver.json 2> gpurun_out/r6c/bench_c2_driver.err
tail -5 gpurun_out/r6c/pytest.log; tail -2 gpurun_out/r6c/bench_share2.err; head -c 300 gpurun_out/r6c/bench_share2.json
