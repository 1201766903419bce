#!/bin/bash
# compiled one-environment online loop (examples/online_loop_atari): device and host observations, B = 32 and 256
mkdir -p gpurun_out
{
  for rep in 1 2; do
    for where in device host; do
      for b in 256 32; do
        examples/online_loop_atari 6000 $b 0 $where | head -1
      done
    done
  done
} > gpurun_out/online_loop_compiled.txt 2>&1
cat gpurun_out/online_loop_compiled.txt
python -m pytest tests/test_gpu_trainer.py -q -m gpu -k "compiled" 2>&1 | tail -3
