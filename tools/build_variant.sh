#!/bin/bash
# build a variant of libborder_amd.so with extra -D flags for dqn.hip: tools/build_variant.sh name -DX=1 ...
name=$1; shift
cd /root/repo/border_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function "$@" -c dqn.hip -o /root/repo/scratch/variants/dqn_$name.o || exit 1
objs=""
for f in replay per agent_api mlp_agents sac iqn comm atari_prep trainer async_trainer; do objs="$objs $f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/lib_$name.so /root/repo/scratch/variants/dqn_$name.o $objs -ldl
