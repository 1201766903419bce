#!/bin/bash
# round 4, device actor path: its tests, then the actor-side probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_device_actor.py tests/test_gpu_atari_prep.py tests/test_gpu_sample.py -m gpu -q -x 2>&1 | tail -25 > $O/r4b_tests.log
tail -4 $O/r4b_tests.log
timeout 600 python tools/probes/online_loop_device.py > $O/r4b_online_loop_device.log 2>&1
cat $O/r4b_online_loop_device.log | tail -20
