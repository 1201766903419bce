#!/bin/bash
# round 6: the bench lines of the evidence pack, re-run AFTER the pack's kernel traces / PMC pass are in profiles/ (same kernel sources), so
# that every line carries `traffic`, `traffic_vs_algorithmic` and `fp32_gemm_sum` of those files
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
tag=${1:-r06}
for c in c2 c4 c5 c1; do timeout 900 python bench.py --config $c > $O/bench_${tag}_$c.json 2> $O/bench_${tag}_$c.err; done
BDR_NO_SAC_FUSE=1 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_layer_by_layer.json 2>/dev/null
BDR_SAC_SIDE_QUEUE=0 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_one_queue.json 2>/dev/null
BDR_NO_SAC_CHAIN=1 BDR_SAC_TAIL_IN_KERNEL=1 BDR_SAC_WAIT_PACKET=1 timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_${tag}_c5_round4_form.json 2>/dev/null
timeout 600 python bench.py --config c2 --frame-ring --no-cpu-baseline > $O/bench_${tag}_c2_frame_ring.json 2>/dev/null
timeout 600 python bench.py --config c2 --per --no-cpu-baseline > $O/bench_${tag}_c2_per.json 2>/dev/null
timeout 600 python bench.py --config c4 --arithmetic f32_exact --no-cpu-baseline > $O/bench_${tag}_c4_exact_f32.json 2>/dev/null
timeout 600 python bench.py --config c2 --arithmetic f32_exact --no-cpu-baseline > $O/bench_${tag}_c2_exact_f32.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_c2_driver_form.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_c2_driver_form_2.json 2>/dev/null
for f in $O/bench_${tag}_*.json; do python - $f <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d["value"], (d.get("cold_window") or {}).get("value"), (d.get("steady_state") or {}).get("value"), r.get("kernel"), r.get("frac"), r.get("traffic"),
          (r.get("fp32_gemm_sum") or {}).get("frac"), (r.get("fp32_gemm_sum") or {}).get("same_kernel_sources"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
