#!/bin/bash
# scratch/ab/libborder_amd_<name>.so: the current objects with dqn.hip recompiled under extra -D flags (same-box A/B of compile-time shapes).
# usage: build_dqn_variant.sh <name> -DBDR_X=... ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$root/scratch/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c "$root/border_amd/csrc/dqn.hip" -o "$root/scratch/ab/dqn_$name.o"
objs=$(ls "$root"/border_amd/csrc/*.o | grep -v -e '/dqn.o$' -e '/comm_hostcomm.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/scratch/ab/libborder_amd_$name.so" $objs "$root/scratch/ab/dqn_$name.o" -ldl
echo "$root/scratch/ab/libborder_amd_$name.so"
