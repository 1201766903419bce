cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "BDR_SAC_CHAIN_TPW=4" "BDR_NO_SAC_CHAIN=1"; do
  env $v python bench.py --config c5 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value']); k=d['roofline']['kernels_ms']; print({a:round(b*1000,2) for a,b in k.items()}); print(sum(k.values())*1000)"
done > gpurun_out/c5k.log 2>&1
cat gpurun_out/c5k.log
