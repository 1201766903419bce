"""Two REAL ranks through csrc/comm.hip with nranks = 2: the N>1 data planes of SURVEY.md 8(e) on hardware.

On a box with >= 2 MI355X: one process per GPU over RCCL.  On the 1-GPU lease RCCL refuses two ranks on one device, so the two ranks
SHARE device 0 and load libborder_amd_hostcomm.so - the same objects with comm.hip compiled under -DBDR_COMM_HOST_TRANSPORT, i.e. the six
librccl entry points replaced by a host shared-memory transport (csrc/comm_host_transport.hpp) and every line ABOVE them unchanged:
bdr_comm_agree, bdr_agent_allreduce_params with its per-segment overlapped exchange, the 1/N scale, bdr_agent_set_grad_comm,
bdr_agent_broadcast_params, the async trainer's exchange / agree hooks.  (Rounds 1-5 skipped all of this at one GPU: comm.hip had never seen
a second rank.)  torch.distributed (gloo) is only the control plane (unique-id hand-off, barriers), exactly as in bench.py.

Parity statements (8(e) "Parity at G>1"), each against something that IS parity-checked on one GPU:
 (i)   synchronous data-parallel, 2 x 128 rows: grads_on_batch -> ncclAllReduce(grad)/2 -> apply_grads on both ranks == the C
       oracle's ONE step on the 256-row batch (gradient and parameters), and the ranks stay bit-identical;
       production form (bdr_agent_set_grad_comm + Agent::opt over identical rings) == the un-exchanged single-rank run, bit for bit
       ((x + x) / 2 == x in f32);
 (ii)  parameter averaging after K local steps == fl(fl(P0 + P1) * 0.5) of the two ranks' own parameters, which equal independent
       single-GPU runs of the same seeds bit for bit;
 (iii) broadcast == the root's parameters, bit for bit (the reference's learner -> actors sync, async_trainer/base.rs:268-272);
 (iv)  the overlapped per-segment exchange (communication queue beside the backward) == the in-stream exchange, bit for bit;
 (v)   bdr_comm_agree: MIN of the ranks' ok flags on every rank;
 (vi)  the SAC handle: SyncModel ships `pi` only (sac/base.rs:377-386) - after K local steps on different shards the exchange leaves
       `pi` == fl(fl(pi0 + pi1) * 0.5) on both ranks and every critic, target critic and log_alpha exactly as it was;
 (vii) the IQN handle: its whole model (`iqn`) is averaged, `iqn_tgt` and the Adam moments stay local;
 (viii) bdr_async_train per rank (learner + actor + local shard) with the `exchange` + `agree` hooks over the two real ranks
       (async_trainer/base.rs:268-272, 299-388): both ranks stop after max_opts, sync the same number of times, and - the run ends with a
       sync - hold bit-identical learner parameters.
The rank script itself also runs with ONE rank on the 1-GPU box (test_rank_script_runs_on_one_rank: every collective is then the
identity), so that what only a 2-GPU node can check is at least free of programming errors before it gets there."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    try:
        import border_amd
        return border_amd.device_count()
    except Exception:  # noqa: BLE001
        return 0


HOSTCOMM = os.path.join(ROOT, "border_amd", "libborder_amd_hostcomm.so")
# two ranks: RCCL over two GPUs when the box has them, else the host-transport build of the communicator with both ranks on device 0
SHARE_ONE_GPU = _n_gpus() < 2
needs_two = pytest.mark.skipif(_n_gpus() < 1 or (SHARE_ONE_GPU and not os.path.exists(HOSTCOMM)),
                               reason="needs an MI355X and, with fewer than two of them, border_amd/libborder_amd_hostcomm.so (build.build_hostcomm_library)")


def _cnn(B, bs, dev, **kw):
    return B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                                   device=dev, batch_size=bs, critic_loss="SmoothL1", **kw))


def _ring(B, dev, seed, fill_seed, n=800):
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=n, seed=seed), (4, 1, 84, 84), np.uint8, device=dev)
    rb.fill_synthetic(n, seed=fill_seed, kind=0, n_actions=6)
    return rb


def _rank_main(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import ctypes as C
    import torch
    import torch.distributed as dist
    import border_amd as B
    from oracle import torch_ref as T
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = B._lib.lib()
    dev = 0 if os.environ.get("BDR_TEST_SHARE_GPU") == "1" else rank

    def bcast_bytes(b):
        t = torch.zeros(B._lib.BDR_UNIQUE_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            t = torch.tensor(list(b), dtype=torch.uint8)
        dist.broadcast(t, src=0)
        return bytes(t.tolist())
    ex = B.ParamExchange.rccl_or_raise(world, rank, 1, dev, bcast_bytes, ("qnet",))
    comm = ex._comm
    own_comm = None
    if world == 1:   # the one-rank run of this script: a real 1-rank RCCL communicator for the direct calls (every collective = identity)
        uid = (C.c_uint8 * B._lib.BDR_UNIQUE_ID_BYTES)()
        B._lib.check(L.bdr_comm_get_unique_id(uid))
        own_comm = C.c_void_p()
        B._lib.check(L.bdr_comm_init_rank(uid, 1, 0, dev, C.byref(own_comm)))
        comm = own_comm
    res = {"rank": rank}

    # (v) agreement
    res["agree_all_ok"] = ex.agree(True)
    res["agree_one_failed"] = ex.agree(rank != 1)

    # (i) synchronous DP on a fixed 256-row batch, halves per rank; three steps
    p0 = T.init_params(T.cnn_shapes(6), 7)
    a = _cnn(B, 128, dev, tau=1.0, soft_update_interval=10000)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    sync = []
    for step in range(3):
        obs, act, nobs, rew, term = T.synthetic_atari_batch(256, 6, 500 + step)
        h = tuple(x[rank * 128:(rank + 1) * 128] for x in (obs, act, nobs, rew, term))
        rec = a.grads_on_batch(*h)
        B._lib.check(L.bdr_agent_allreduce_params(a.handle, comm, a.WHICH["grad"]))     # sum over ranks, then 1/2
        g = a.get_params("grad")
        a.apply_grads()
        sync.append({"loss": rec["loss"], "grad": g, "qnet": a.get_params("qnet"), "n_opts": a.n_opts})
    res["sync_dp"] = sync
    a.close()

    # (i') production form: set_grad_comm + opt over IDENTICAL rings and seeds on both ranks == the run without a communicator
    outs = []
    for use_comm in (False, True):
        rb = _ring(B, dev, 42, 3, 600)
        a = _cnn(B, 32, dev, tau=0.5, soft_update_interval=4, param_seed=9)
        if use_comm:
            B._lib.check(L.bdr_agent_set_grad_comm(a.handle, comm))
        for _ in range(9):
            a.opt(rb)
        rec = a.opt_with_record(rb)
        outs.append((a.get_params("qnet"), a.get_params("qnet_tgt"), rec["loss"]))
        if use_comm:
            B._lib.check(L.bdr_agent_set_grad_comm(a.handle, None))
        a.close(); rb.close()
    res["grad_comm_identity"] = bool((outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all() and outs[0][2] == outs[1][2])
    res["grad_comm_qnet"] = outs[1][0]

    # (ii) + (iv): K local steps on DIFFERENT shards (seed + rank), then averaging; overlapped and in-stream exchange
    def local_run(exchange_every, overlap, steps=12):
        if overlap:
            os.environ.pop("BDR_NO_XCHG_OVERLAP", None)
        else:
            os.environ["BDR_NO_XCHG_OVERLAP"] = "1"
        rb = _ring(B, dev, B.shard_seed(42, rank), 10 + rank)
        a = _cnn(B, 32, dev, tau=0.5, soft_update_interval=5, param_seed=9)
        before_avg = None
        for s in range(1, steps + 1):
            a.opt(rb)
            if exchange_every and s % exchange_every == 0:
                if s == steps:
                    before_avg = a.get_params("qnet")
                B._lib.check(L.bdr_agent_allreduce_params(a.handle, comm, 0))
        a.sync()
        out = (a.get_params("qnet"), a.get_params("qnet_tgt"), a.get_params("exp_avg"), before_avg)
        a.close(); rb.close()
        os.environ.pop("BDR_NO_XCHG_OVERLAP", None)
        return out
    solo = local_run(0, True)                  # an independent single-GPU run of this rank's shard
    once = local_run(12, True)                 # the same 12 local steps, then ONE average
    res["avg_own_before"] = once[3]
    res["avg_own_equals_solo"] = bool((once[3] == solo[0]).all())
    res["avg_after"] = once[0]
    res["avg_moments_local"] = bool((once[2] == solo[2]).all())      # Adam moments stay local
    ov = local_run(3, True)
    ins = local_run(3, False)
    res["overlap_equals_instream"] = bool((ov[0] == ins[0]).all() and (ov[1] == ins[1]).all() and (ov[2] == ins[2]).all())
    res["exchanged_every_3"] = ov[0]

    # (vi) SAC: only `pi` crosses the ranks
    sac_rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=2000, seed=B.shard_seed(42, rank)), (17,), np.float32, (6,), np.float32, device=dev)
    sac_rb.fill_synthetic(2000, seed=20 + rank, kind=1, n_actions=0)
    sac = B.Sac.build(B.SacConfig(obs_dim=17, act_dim=6, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=64,
                                  ent_coef_mode=("Auto", -6.0, 3e-4), lr_actor=3e-4, lr_critic=3e-4, device=dev, seed=5))
    for _ in range(6):
        sac.opt(sac_rb)
    sac.sync()
    names = ["pi", "qnet_0", "qnet_1", "qnet_tgt_0", "qnet_tgt_1", "log_alpha"]
    before = {n: sac.get_params(n) for n in names}
    B._lib.check(L.bdr_agent_allreduce_params(sac.handle, comm, sac.WHICH["pi"]))   # (the process's one communicator)
    sac.sync()
    res["sac_before"], res["sac_after"] = before, {n: sac.get_params(n) for n in names}
    sac.opt(sac_rb); sac.sync()          # the step after an exchange runs (one-queue / two-queue hand-over)
    res["sac_n_opts"] = sac.n_opts
    sac.close(); sac_rb.close()

    # (vii) IQN: the whole model is averaged, target and moments stay local
    irb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=500, seed=B.shard_seed(42, rank)), (4,), np.float32, device=dev)
    irb.fill_synthetic(500, seed=30 + rank, kind=1, n_actions=3)
    iqn = B.Iqn.build(B.IqnConfig(f_config=B.MlpConfig(in_dim=4, units=(32,), out_dim=16, activation_out=True), feature_dim=16, embed_dim=8, m_units=(32,),
                                  n_actions=3, lr=1e-3, batch_size=16, device=dev, seed=3, tau=0.5, soft_update_interval=3))
    for _ in range(5):
        iqn.opt(irb)
    iqn.sync()
    ib = {n: iqn.get_params(n) for n in ("iqn", "iqn_tgt", "exp_avg")}
    B._lib.check(L.bdr_agent_allreduce_params(iqn.handle, comm, iqn.WHICH["iqn"]))
    iqn.sync()
    res["iqn_before"], res["iqn_after"] = ib, {n: iqn.get_params(n) for n in ("iqn", "iqn_tgt", "exp_avg")}
    iqn.close(); irb.close()

    # (viii) the async trainer on every rank, learners averaged at every sync point, agreement before every collective
    arb = _ring(B, dev, B.shard_seed(42, rank), 40 + rank, 400)
    learner = _cnn(B, 16, dev, tau=1.0, soft_update_interval=10000, param_seed=9)
    actor = _cnn(B, 16, dev, param_seed=9)
    actor.set_explorer(B.EpsilonGreedy(final_step=100), seed=50 + rank)
    env = B.SyntheticEnv((4, 1, 84, 84), np.uint8, seed=60 + rank, p_term=0.05)
    ex.which = ("qnet",)
    events = []
    tr = B.AsyncTrainer(B.AsyncTrainerConfig(max_opts=7, warmup_period=0, sync_interval=3, record_agent_info_interval=0, record_compute_cost_interval=0,
                                             warmup_sleep_ms=1), B.ActorManagerConfig(n_buffer=4))
    st = tr.train(learner, arb, [actor], [env], (4, 1, 84, 84), np.uint8, exchange=lambda s: ex.average(learner), agree=lambda ok: ex.agree(ok),
                  on_event=lambda actor_id, a_, b_, ev, v: events.append((ev, b_)) if ev == "sync" else None)
    learner.sync()
    res["async_stat"] = (st.opt_steps, st.n_syncs, [b_ for ev, b_ in events])
    res["async_qnet"] = learner.get_params("qnet")
    learner.close(); actor.close(); arb.close()

    # (iii) broadcast from rank 1 (rank 0 when there is one rank)
    a = _cnn(B, 32, dev, tau=1.0, soft_update_interval=10000, param_seed=100 + rank)
    mine = a.get_params("qnet")
    B._lib.check(L.bdr_agent_broadcast_params(a.handle, comm, 0, world - 1))
    a.sync()
    res["bcast_mine"], res["bcast_after"] = mine, a.get_params("qnet")
    a.close()

    ex.close()
    if own_comm is not None:
        B._lib.check(L.bdr_comm_destroy(own_comm))
    out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    saved = {k: os.environ.get(k) for k in ("BORDER_AMD_LIB", "BDR_TEST_SHARE_GPU")}
    if world > 1 and SHARE_ONE_GPU:   # (spawned children inherit the environment at start())
        os.environ["BORDER_AMD_LIB"], os.environ["BDR_TEST_SHARE_GPU"] = HOSTCOMM, "1"
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.fixture(scope="module")
def two_ranks():
    return _spawn(2)


def test_rank_script_runs_on_one_rank():
    """The whole rank script over a 1-rank RCCL communicator (runs on the 1-GPU box): every collective is the identity, so each
    'after' equals its 'before' bit for bit, the async trainer finishes with its sync points, and nothing in the script that only
    a 2-GPU node exercises is left untried for plain programming errors."""
    if _n_gpus() < 1:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    (r,) = _spawn(1)
    assert r["agree_all_ok"] is True and r["agree_one_failed"] is True          # (rank 0 never says "failed" here)
    assert r["grad_comm_identity"] and r["avg_own_equals_solo"] and r["avg_moments_local"] and r["overlap_equals_instream"]
    assert (r["avg_after"] == r["avg_own_before"]).all() and (r["bcast_after"] == r["bcast_mine"]).all()
    for n, v in r["sac_before"].items():
        assert (r["sac_after"][n] == v).all(), n
    assert r["sac_n_opts"] == 7
    for n, v in r["iqn_before"].items():
        assert (r["iqn_after"][n] == v).all(), n
    opt_steps, n_syncs, at = r["async_stat"]
    assert opt_steps == 7 and at == [0, 3, 6, 7] and n_syncs == 4                # first, every sync_interval, last (util.rs:31-92)


@needs_two
def test_agreement_is_the_min_over_ranks(two_ranks):
    assert all(r["agree_all_ok"] is True for r in two_ranks)
    assert all(r["agree_one_failed"] is False for r in two_ranks)


@needs_two
def test_sync_dp_two_ranks_of_128_equal_the_oracles_step_on_256(two_ranks):
    from oracle import oracle as O
    from oracle import torch_ref as T
    from tests.test_gpu_dqn import assert_grads_close
    shapes = T.cnn_shapes(6)
    p0 = T.init_params(shapes, 7)
    ref = O.DqnOracle(O.cnn_cfg(6), p0, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000)
    r0, r1 = two_ranks[0]["sync_dp"], two_ranks[1]["sync_dp"]
    for step in range(3):
        assert (r0[step]["grad"] == r1[step]["grad"]).all() and (r0[step]["qnet"] == r1[step]["qnet"]).all()    # lock step, bit for bit
        assert r0[step]["n_opts"] == r1[step]["n_opts"] == step + 1
    obs, act, nobs, rew, term = T.synthetic_atari_batch(256, 6, 500)
    r = ref.update(obs, act, nobs, rew, term, probe=True)
    assert_grads_close(r0[0]["grad"], r["grads"], shapes)                                   # all-reduced mean gradient == full-batch gradient
    assert abs(0.5 * (r0[0]["loss"] + r1[0]["loss"]) - r["loss"]) <= 1e-4 * abs(r["loss"])
    dp = np.abs(r0[0]["qnet"].astype(np.float64) - ref.q)
    assert (dp > 0.05 * 1e-4).sum() <= 2 * 257 and dp.max() <= 2.5e-4, ((dp > 0.05 * 1e-4).sum(), dp.max())


@needs_two
def test_grad_comm_over_identical_shards_is_the_identity(two_ranks):
    assert all(r["grad_comm_identity"] for r in two_ranks)
    assert (two_ranks[0]["grad_comm_qnet"] == two_ranks[1]["grad_comm_qnet"]).all()


@needs_two
def test_parameter_average_is_the_mean_of_two_independent_runs(two_ranks):
    assert all(r["avg_own_equals_solo"] and r["avg_moments_local"] for r in two_ranks)
    p0, p1 = two_ranks[0]["avg_own_before"], two_ranks[1]["avg_own_before"]
    assert not (p0 == p1).all()                                   # different shards really diverged
    mean = (p0 + p1) * np.float32(0.5)                            # f32 sum (commutative for two ranks), then the 1/N scale
    assert (two_ranks[0]["avg_after"] == mean).all() and (two_ranks[1]["avg_after"] == mean).all()


@needs_two
def test_overlapped_exchange_equals_the_in_stream_exchange(two_ranks):
    assert all(r["overlap_equals_instream"] for r in two_ranks)
    assert (two_ranks[0]["exchanged_every_3"] == two_ranks[1]["exchanged_every_3"]).all()     # step 12 ended with an average


@needs_two
def test_broadcast_equals_the_root(two_ranks):
    root = two_ranks[1]["bcast_mine"]
    assert not (two_ranks[0]["bcast_mine"] == root).all()
    assert (two_ranks[0]["bcast_after"] == root).all() and (two_ranks[1]["bcast_after"] == root).all()


@needs_two
def test_sac_exchange_ships_pi_only(two_ranks):
    b0, b1 = two_ranks[0]["sac_before"], two_ranks[1]["sac_before"]
    assert not (b0["pi"] == b1["pi"]).all()                                       # different shards diverged
    mean = (b0["pi"] + b1["pi"]) * np.float32(0.5)
    for r in two_ranks:
        assert (r["sac_after"]["pi"] == mean).all()
        for n in ("qnet_0", "qnet_1", "qnet_tgt_0", "qnet_tgt_1", "log_alpha"):   # sac/base.rs:377-386: the critics and alpha stay local
            assert (r["sac_after"][n] == r["sac_before"][n]).all(), n
        assert r["sac_n_opts"] == 7


@needs_two
def test_iqn_exchange_averages_the_model_and_leaves_target_and_moments_local(two_ranks):
    b0, b1 = two_ranks[0]["iqn_before"], two_ranks[1]["iqn_before"]
    assert not (b0["iqn"] == b1["iqn"]).all()
    mean = (b0["iqn"] + b1["iqn"]) * np.float32(0.5)
    for r in two_ranks:
        assert (r["iqn_after"]["iqn"] == mean).all()
        assert (r["iqn_after"]["iqn_tgt"] == r["iqn_before"]["iqn_tgt"]).all() and (r["iqn_after"]["exp_avg"] == r["iqn_before"]["exp_avg"]).all()


@needs_two
def test_async_trainers_on_two_ranks_average_at_every_sync_point_and_end_identical(two_ranks):
    s0, s1 = two_ranks[0]["async_stat"], two_ranks[1]["async_stat"]
    assert s0 == s1 and s0[0] == 7 and s0[2] == [0, 3, 6, 7] and s0[1] == 4       # same sync points on both ranks (agree + exchange at each)
    assert (two_ranks[0]["async_qnet"] == two_ranks[1]["async_qnet"]).all()        # the run ends with a sync: averaged learners
