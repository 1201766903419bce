#!/usr/bin/env python3
"""Headline benchmark: agent opt-steps/sec, DQN Atari 84x84x4, batch 256 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one Agent::opt() -- on-device ChaCha12 index draw + gather from the 1M-transition HBM
ring + online/target forward + Huber TD loss + backward + Adam (+ target sync when due) -- exactly
the region Trainer::train_step times (border-core/src/trainer.rs:213-225).  Inputs are resident in
HBM before the timed region starts.  For N>1 the driver launches one rank per GPU with
torch.distributed.run; every rank owns a replica + a local replay shard and parameters are averaged
over RCCL every --sync-interval opt steps (inside the timed region).

Prints ONE JSON line (rank 0).  `roofline` is measured live: per-kernel HIP-event timing on the
agent's own stream (bdr_agent_profile_*), algorithmic FLOPs from SURVEY.md section 8(d).
`cpu_baseline` times oracle/torch_ref.py (the ATen op sequence tch 0.16 binds) on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ACTIONS = 6
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def kernel_flops(B, nz):
    """Algorithmic FLOPs per launch (2*MACs), SURVEY.md section 8(d) / section 2.3 shapes."""
    c1, c2, c3, l1 = 2 * B * 400 * 256 * 32, 2 * B * 81 * 512 * 64, 2 * B * 49 * 576 * 64, 2 * B * 3136 * 512
    return {"fwd_conv1": nz * c1, "fwd_conv2": nz * c2, "fwd_conv3": nz * c3, "fwd_l1": nz * l1,
            "bwd_conv1_dw": c1, "bwd_conv2_dw": c2, "bwd_conv2_dx": c2, "bwd_conv3_dw": c3, "bwd_conv3_dx": c3,
            "bwd_l1_dw": l1, "bwd_l1_dx": l1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--capacity", type=int, default=1_000_000)
    ap.add_argument("--loss", default="SmoothL1", choices=["SmoothL1", "Mse"])
    ap.add_argument("--double-dqn", action="store_true")
    ap.add_argument("--sync-interval", type=int, default=10, help="opt steps between RCCL parameter averaging (N>1)")
    ap.add_argument("--per", action="store_true", help="prioritized replay (PerConfig defaults) instead of uniform sampling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--profile-steps", type=int, default=30)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch  # noqa: F401  (first: one HIP runtime per process, see border_amd/_lib.py)
    import border_amd as B

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barrier, id hand-off, max-reduce of the timing) on gloo; the data plane
        # (parameter all-reduce) is the library's own RCCL communicator over xGMI
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if os.environ.get("BDR_BENCH_SHARE_GPU") == "1":   # flow test of the N>1 path on a 1-GPU box (ranks share device 0)
        local_rank = local_rank % max(B.device_count(), 1)
    if B.device_count() <= local_rank:
        sys.exit(f"rank {rank}: HIP device {local_rank} not visible")

    def bcast_bytes(b):
        t = torch.zeros(B._lib.BDR_UNIQUE_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            t = torch.tensor(list(b), dtype=torch.uint8)
        dist.broadcast(t, src=0)
        return bytes(t.tolist())

    # replay shard: 1M transitions of synthetic 84x84x4 u8 frames per GPU, own StdRng stream
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=args.capacity, seed=B.shard_seed(42, rank),
                                                          per_config=B.PerConfig() if args.per else None),
                              (4, 1, 84, 84), "uint8", device=local_rank)
    rb.fill_synthetic(args.capacity, seed=rank, kind=0, n_actions=N_ACTIONS)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=N_ACTIONS),
                                                    opt_config=B.OptimizerConfig.Adam(1e-4)),
                      soft_update_interval=10000, n_updates_per_opt=1, batch_size=args.batch, discount_factor=0.99,
                      tau=1.0, double_dqn=args.double_dqn, critic_loss=args.loss, device=local_rank, param_seed=0)
    agent = B.Dqn.build(cfg)
    agent.train()
    # data plane: the library's own RCCL communicator; demoted (by all ranks together) to torch's RCCL on the same
    # device arena, then to host staging, only if the communicator cannot be brought up on this node
    exch = B.ParamExchange.with_fallback(world, rank, args.sync_interval, local_rank, bcast_bytes) if world > 1 else None

    def run(n, first_step):
        for s in range(n):
            agent.opt(rb)
            if exch is not None:
                exch.after_opt(agent, first_step + s + 1)

    run(args.warmup, 0)
    agent.sync()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    agent.sync()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    result = None
    if rank == 0:
        ms = 1000.0 * dt / args.steps
        value = world * args.steps / dt
        # roofline leg: per-kernel HIP-event timing on the agent's stream
        # An empty event bracket measures TWO marker packets back to back; a bracket around a kernel contains the
        # kernel plus ONE marker's processing time (the closing marker is stamped when the kernel retires).  So
        # the per-kernel correction is half of the empty bracket; this reproduces rocprofv3's kernel durations
        # to ~0.3 us (profiles/rocprof_r01_kernel_trace_v4.md).
        # the timed steps trained a real network: one more step with its Record, which must be finite
        rec = agent.opt_with_record(rb)
        final_loss = float(rec["loss"])
        if not (final_loss == final_loss and abs(final_loss) != float("inf")):
            sys.exit(f"bench.py: non-finite loss {final_loss} after the timed steps")

        def profile(n):
            agent.profile_enable(True)
            for _ in range(n):
                agent.opt(rb)
            p = agent.profile_read()
            agent.profile_enable(False)
            half_null = 0.5 * p.pop("_null", 0.0)
            return {k: max(v - half_null, 0.0) for k, v in p.items()}, half_null
        prof, null_ms = profile(args.profile_steps)
        if any(v <= 0.0 for v in prof.values()):   # an outlier empty bracket (its half is subtracted everywhere): measure again
            prof, null_ms = profile(max(30, args.profile_steps))
        nz = 3 if args.double_dqn else 2
        fl = kernel_flops(args.batch, nz)
        if not any(k in fl for k in prof):
            sys.exit("bench.py needs --profile-steps >= 1 for the roofline leg")
        dom = max((k for k in prof if k in fl), key=lambda k: prof[k])
        achieved = fl[dom] / (max(prof[dom], 1e-6) * 1e-3) / 1e12
        step_flops = sum(fl.values()) + 2 * nz * args.batch * 512 * N_ACTIONS
        gather_bytes = 2 * args.batch * 28224 + args.batch * 14
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                "kernel_ms": round(prof[dom], 5),
                "step": {"gflop": round(step_flops / 1e9, 3), "achieved": round(step_flops / (ms * 1e-3) / 1e12, 2),
                         "frac": round(step_flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
                "gather": {"bound": "hbm", "bytes": gather_bytes, "ms": round(prof.get("sample", 0.0), 5),
                           "achieved_GBs": round(gather_bytes / max(prof.get("sample", 1e9), 1e-9) / 1e6, 1),
                           "peak_GBs": PEAK_HBM_GBS},
                "event_bracket_overhead_ms": round(null_ms, 5),
                "kernels_ms": {k: round(v, 5) for k, v in prof.items()}}
        tr = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tr):   # PMC-derived HBM bytes per launch, measured with rocprofv3 --pmc (profiles/)
            try:
                roof["traffic"] = json.load(open(tr)).get(dom)
            except Exception:
                pass
        result = {"metric": "agent opt-steps/sec (DQN Atari 84x84x4, batch 256)", "value": round(value, 2),
                  "unit": "opt-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"synthetic Atari DQN Nature-CNN, replay {args.capacity} u8 transitions/GPU, batch {args.batch}/GPU",
                             "batch_size": args.batch, "replay_capacity": args.capacity, "n_actions": N_ACTIONS,
                             "critic_loss": args.loss, "double_dqn": args.double_dqn, "prioritized_replay": bool(args.per),
                             "optimizer": "Adam lr=1e-4",
                             "soft_update_interval": 10000, "tau": 1.0,
                             "parallelism": f"dp{world} (replica + replay shard per GPU"
                                            + (f", parameter all-reduce every {args.sync_interval} opts over {exch.backend})" if world > 1 else ")"),
                             "final_loss": round(final_loss, 6), "samples_per_sec": round(value * args.batch, 1)},
                  "roofline": roof}
    agent.close()
    rb.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the reference's ATen op sequence on the host cores (bounded sample)
        from oracle import torch_ref
        sps, threads = torch_ref.time_dqn_atari(args.batch, N_ACTIONS, steps=2, warmup=1, critic_loss=args.loss)
        n = max(3, min(400, int(args.cpu_seconds * sps)))
        sps, threads = torch_ref.time_dqn_atari(args.batch, N_ACTIONS, steps=n, warmup=1, critic_loss=args.loss)
        result["cpu_baseline"] = {"value": round(sps, 3), "unit": "opt-steps/s", "cores": threads, "kind": "port",
                                  "sample": f"{n} opt steps (batch {args.batch}, f32 ring of 4096 transitions) of "
                                            "oracle/torch_ref.py: the libtorch-CPU op sequence of border-tch-agent",
                                  "gpu_over_cpu": round(result["value"] / sps, 1)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if exch is not None:
        exch.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
