#!/bin/bash
# Regenerates the libtorch-written checkpoint fixtures (tests/test_checkpoint_archive.py) with the libtorch inside the
# installed PyTorch wheel, through the call sequence tch's `VarStore::save` uses (oracle/libtorch_archive.cpp).
# Tensor values are sin(0.37 * (element + 1) + tensor_index), so the test recomputes the expected contents itself.
set -e
cd "$(dirname "$0")/../.."
bash oracle/build_libtorch_archive.sh
T=oracle/_build/libtorch_archive
# Mlp(4,[64,64],2) (border-tch-agent/src/mlp/base.rs:13-41 names), the CartPole Q-net of BASELINE config 1
$T write tests/golden/libtorch_mlp_qnet.pt.tch mlp.ln0.weight 64,4 mlp.ln0.bias 64 mlp.ln1.weight 64,64 mlp.ln1.bias 64 mlp.ln2.weight 2,64 mlp.ln2.bias 2
# AtariCnn names (cnn/base.rs:23-36) with a 1-d scalar-like and a 4-d tensor, small shapes
$T write tests/golden/libtorch_small_cnn.pt.tch c1.weight 3,4,8,8 c1.bias 3 l2.weight 6,5 l2.bias 6 log_alpha 1
ls -l tests/golden/*.pt.tch
