cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 4 8; do
  BDR_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus $n --steps 40 --warmup 5 --capacity 20000 > gpurun_out/bench_share$n.json 2> gpurun_out/bench_share$n.err
  echo "share$n rc=$?"; cut -c1-260 gpurun_out/bench_share$n.json; tail -2 gpurun_out/bench_share$n.err
done
# the driver's N>1 form: torch.distributed.run wraps bench.py
BDR_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --capacity 20000 > gpurun_out/bench_tdr2.json 2> gpurun_out/bench_tdr2.err
echo "tdr2 rc=$?"; cut -c1-200 gpurun_out/bench_tdr2.json; wc -l gpurun_out/bench_tdr2.json
