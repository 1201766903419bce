"""N>1 path on CPU: world_size-2 gloo processes drive border_amd.ParamExchange's schedule (after_opt intervals, the RCCL-or-nothing
ladder agreed on by every rank) and the shard/seed logic (no GPU in this container).  The product class has ONE data plane, the
library's communicator, which needs a GPU: here a TEST-SIDE subclass (GlooExchange, below) moves the vectors over gloo instead.  The
real call sequences of csrc/comm.hip with two ranks run in tests/test_gpu_multi.py (-m gpu; on a 1-GPU box over the communicator's
host-transport test build).
Properties: averaging identical replicas is the identity; the average is
the arithmetic mean of the ranks' parameters; broadcast makes every rank equal to the root;
shards draw from distinct StdRng streams."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeAgent:
    """Implements the slice of the Dqn mirror that ParamExchange uses (get/set_params, WHICH)."""
    WHICH = {"qnet": 0, "qnet_tgt": 1}

    def __init__(self, p):
        self.p = {"qnet": np.array(p, np.float32), "qnet_tgt": np.array(p, np.float32)}
        self.handle = None

    def get_params(self, which="qnet"):
        return self.p[which]

    def set_params(self, v, which="qnet"):
        self.p[which] = np.array(v, np.float32)


def _gloo_exchange_class():
    """ParamExchange with its three collectives on torch.distributed / gloo host vectors: the test's own data plane (the product has none
    but the library's communicator)."""
    import torch
    import torch.distributed as dist
    from border_amd.trainer import ParamExchange

    class GlooExchange(ParamExchange):
        def _connect(self, bcast_bytes):
            return None

        def average(self, agent):
            for w in self.which:
                t = torch.from_numpy(np.array(agent.get_params(w), copy=True))
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t /= self.world_size
                agent.set_params(t.numpy(), w)

        def agree(self, local_ok=True):
            t = torch.tensor([1 if local_ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t[0]))

        def broadcast(self, agent, root=0):
            for w in self.which:
                t = torch.from_numpy(np.array(agent.get_params(w), copy=True))
                dist.broadcast(t, src=root)
                agent.set_params(t.numpy(), w)

    return GlooExchange


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from border_amd.trainer import ParamExchange, shard_seed
    from oracle import oracle as O
    ex = _gloo_exchange_class()(world, rank, sync_interval=3)
    ok_agree = ex.agree(True) is True and ex.agree(rank != 1) is False
    base = np.linspace(-1, 1, 1000).astype(np.float32)
    # (i) identical replicas: averaging is the identity
    a = FakeAgent(base)
    ex.average(a)
    ok_identity = bool((a.get_params() == base).all())
    # (ii) mean of different replicas; only every sync_interval-th step exchanges
    b = FakeAgent(base * (rank + 1))
    fired = [ex.after_opt(b, s) for s in (1, 2, 3)]
    expect = base * np.float32(sum(range(1, world + 1))) / np.float32(world)
    ok_mean = bool(np.allclose(b.get_params(), expect, rtol=1e-6, atol=1e-7)) and fired == [False, False, True]
    # (iii) broadcast from the root
    c = FakeAgent(base + rank)
    ex.broadcast(c, root=0)
    ok_bcast = bool((c.get_params() == base).all())
    # (v) RCCL or nothing: without a GPU the communicator cannot come up, and EVERY rank must get the error together
    #     (no rank left in a collective, no silent demotion to a slower data plane)
    try:
        ParamExchange.rccl_or_raise(world, rank, 3, 0, lambda b: b if b is not None else bytes(128))
        ok_ladder = False
    except RuntimeError as e:
        ok_ladder = "RCCL communicator could not be initialised" in str(e)
    # (iv) distinct replay streams per shard
    ix = O.StdRng.seed_from_u64(shard_seed(42, rank)).sample_indices(1_000_000, 8).tolist()
    out.put((rank, ok_identity, ok_mean, ok_bcast and ok_ladder and ok_agree, ix))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=420) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_identity, ok_mean, ok_bcast, ix in res:
        assert ok_identity and ok_mean and ok_bcast, (rank, ok_identity, ok_mean, ok_bcast)
    assert res[0][4] != res[1][4]
