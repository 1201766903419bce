#!/bin/bash
# round 3, call A: all GPU tests, the self-launching N=2 bench on the 1-GPU lease (share mode), the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/tests_gpu.log
echo "rc=${PIPESTATUS[0]}" >> gpurun_out/tests_gpu.log
BDR_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 200 --warmup 20 --capacity 100000 > gpurun_out/bench_share2.json 2> gpurun_out/bench_share2.err
echo "share2 rc=$?" >> gpurun_out/tests_gpu.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_form.json 2> gpurun_out/bench_driver_form.err
echo "bench rc=$?" >> gpurun_out/tests_gpu.log
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 600 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
tail -8 gpurun_out/tests_gpu.log; cat gpurun_out/bench_share2.json | cut -c1-600; tail -3 gpurun_out/bench_share2.err; cut -c1-400 gpurun_out/bench_c2.json; cut -c1-300 gpurun_out/bench_c5.json
