#!/bin/bash
# Builds oracle/_build/libtorch_archive (test infrastructure) against the libtorch shipped inside the installed PyTorch wheel.
set -e
cd "$(dirname "$0")"
TP=$(python -c 'import torch, os; print(os.path.dirname(torch.__file__))')
mkdir -p _build
g++ -O1 -std=c++17 libtorch_archive.cpp -o _build/libtorch_archive \
    -I"$TP/include" -I"$TP/include/torch/csrc/api/include" -L"$TP/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TP/lib"
echo oracle/_build/libtorch_archive
