#!/bin/bash
# variant correctness fingerprints (same seeded opt stream on every library), then the A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6e
for v in "" fwdc2b3_21 fwdc2b3_41 fwdc3b3_21 fwdc3b3_41 fwdc2b3_2112 fwdc2b3_1122 dxc3_t1; do
  if [ -z "$v" ]; then echo -n "[default] "; python tools/probes/variant_check.py 2>/dev/null | tail -1
  else echo -n "[$v] "; BORDER_AMD_LIB=$PWD/scratch/ab/libborder_amd_$v.so python tools/probes/variant_check.py 2>/dev/null | tail -1; fi
done > gpurun_out/r6e/fingerprints.txt
cat gpurun_out/r6e/fingerprints.txt
