#!/bin/bash
# Builds scratch/ab/libborder_amd_prev.so from the csrc/ of a git revision (default HEAD) for same-box A/B runs
# (tools/probes/ab_libs.sh).  usage: build_prev.sh [rev]
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d /tmp/bdr_prev.XXXX)
git -C "$root" archive "$rev" border_amd/csrc include | tar -x -C "$tmp"
objs=""
for f in "$tmp"/border_amd/csrc/*.hip; do
    o="${f%.hip}.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -c "$f" -o "$o" &
    objs="$objs $o"
done
wait
mkdir -p "$root/scratch/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/scratch/ab/libborder_amd_prev.so" $objs -ldl
rm -rf "$tmp"
echo "$root/scratch/ab/libborder_amd_prev.so"
