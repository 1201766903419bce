#!/bin/bash
# the committed bench lines of the round (profiles/bench_<tag>_*.json), run with profiles/hbm_traffic.json + kernel traces of the same sources in place
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
tag=${1:-r03}
for c in c2 c4 c5 c1; do
  timeout 900 python bench.py --config $c > $O/bench_${tag}_$c.json 2> $O/bench_${tag}_$c.err
done
timeout 600 python bench.py --config c2 --frame-ring --no-cpu-baseline > $O/bench_${tag}_c2_frame_ring.json 2>/dev/null
timeout 600 python bench.py --config c2 --per --no-cpu-baseline > $O/bench_${tag}_c2_per.json 2>/dev/null
BDR_NO_SAC_FUSE=1 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_layer_by_layer.json 2>/dev/null
BDR_SAC_SIDE_QUEUE=0 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_one_queue.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_c2_driver_form.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_c2_driver_form_2.json 2>/dev/null
ls $O | grep bench_${tag}
