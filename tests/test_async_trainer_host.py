"""The compiled async-trainer loops (csrc/async_trainer.hip) against the rules of border-async-trainer
(async_trainer/base.rs:204-222, 268-284, 299-388; actor/base.rs:120-178; replay_buffer_proxy.rs:52-72), driven with mock
learner / actors / environments / mailbox callbacks - no GPU involved; plus the world-size-2 gloo run of the same loop with the
cross-rank `exchange` hook (one learner per rank, parameters averaged at every sync point)."""
import ctypes as C
import os
import socket
import sys
import threading
import time

import numpy as np
import pytest
import torch.multiprocessing as mp

from border_amd import _lib
from border_amd.async_trainer import (ActorManagerConfig, ActorOps, AsyncTrainer, AsyncTrainerConfig, LearnerOps, LEN_FN, PUBLISH_FN, SYNC_FN,
                                      env_vtable)
from border_amd.trainer import Step

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class MockEnv:
    """obs = [id*1e6 + t] * 4 (exact in f32 for the step counts reached here); done every `period` steps."""
    def __init__(self, ident, period=7):
        self.id, self.t, self.period = ident, 0, period

    def _obs(self):
        self.t += 1
        return np.full((1, 4), self.id * 1000000 + self.t, np.float32)

    def reset(self, _=None):
        return self._obs()

    def step_with_reset(self, act):
        obs = self._obs()
        done = (self.t % self.period) == 0
        st = Step(np.asarray(act), obs, np.array([float(self.t)], np.float32), np.array([1 if done else 0], np.int8), np.array([0], np.int8))
        if done:
            st.init_obs = self.reset()
        return st


class Mock:
    """Learner (agent + buffer + mailbox) and actor agents as plain Python state behind the function tables."""
    def __init__(self, n_actors, push_delay=0.0, opt_delay=0.0):
        self.lock = threading.Lock()
        self.log, self.pushed, self.len = [], [], 0
        self.mail = (None, None)                     # (n_opts, payload)
        self.actor_version = [None] * n_actors
        self.actor_samples = [[] for _ in range(n_actors)]   # model version used by every Policy::sample
        self.push_delay, self.opt_delay = push_delay, opt_delay
        self.params = 0.0
        self.keep = []

    def learner_ops(self):
        def set_train(_a, on):
            self.log.append(("train", on)); return 0

        def opt(_a, _b):
            if self.opt_delay:
                time.sleep(self.opt_delay)
            self.params += 1.0
            self.log.append(("opt", self.len)); return 0

        def opt_rec(_a, _b, out, cap, n):
            self.params += 1.0
            self.log.append(("opt_rec", self.len)); out[0] = 0.5; n[0] = 1; return 0

        def push(_b, n, obs, act, nobs, rew, term, trunc):
            if self.push_delay:
                time.sleep(self.push_delay)
            o = np.frombuffer((C.c_char * (16 * n)).from_address(obs), np.float32).reshape(n, 4)[:, 0].copy()
            x = np.frombuffer((C.c_char * (16 * n)).from_address(nobs), np.float32).reshape(n, 4)[:, 0].copy()
            t = np.frombuffer((C.c_char * n).from_address(C.addressof(term.contents)), np.int8).copy()
            self.pushed.append((o, x, t))
            self.len += n
            self.log.append(("push", int(n))); return 0

        def blen(_b, out):
            out[0] = self.len; return 0

        def publish(_a, _m, n_opts):
            with self.lock:
                self.mail = (int(n_opts), self.params)
            self.log.append(("publish", int(n_opts))); return 0

        fns = (_lib.SET_TRAIN_FN(set_train), _lib.SAMPLE_FN(lambda *a: 1), _lib.OPT_FN(opt), _lib.OPT_REC_FN(opt_rec), _lib.PUSH_FN(push),
               LEN_FN(blen), PUBLISH_FN(publish))
        self.keep.append(fns)
        ops = LearnerOps()
        ops.t = _lib.TrainerOps(None, None, *fns[:5])
        ops.buffer_len, ops.publish_model = fns[5], fns[6]
        return ops

    def actor_ops(self, i, env):
        def set_train(_a, on):
            return 0

        def sample(_a, n, obs, act_out):
            self.actor_samples[i].append(self.actor_version[i])
            C.cast(act_out, C.POINTER(C.c_int64))[0] = len(self.actor_samples[i])
            return 0

        def sync(_a, _m, actor_id, first, n_opts, updated):
            assert actor_id == i
            with self.lock:
                v, _ = self.mail
            if v is None:
                return 7
            if first or v > n_opts[0]:
                self.actor_version[i] = v
                n_opts[0] = v
                if updated:
                    updated[0] = 1
            return 0

        fns = (_lib.SET_TRAIN_FN(set_train), _lib.SAMPLE_FN(sample), SYNC_FN(sync))
        self.keep.append(fns)
        ops = ActorOps()
        ops.agent_set_train, ops.agent_sample, ops.sync_model = fns
        ops.env = env_vtable(env, (4,), np.float32, keep=self.keep)
        return ops


def run(cfg, man, n_actors=2, **mock_kw):
    m = Mock(n_actors, **mock_kw)
    envs = [MockEnv(i + 1) for i in range(n_actors)]
    events = []
    tr = AsyncTrainer(cfg, man)
    tr.train(None, None, [None] * n_actors, envs, (4,), np.float32, on_event=lambda *e: events.append(e),
             learner_ops=m.learner_ops(), actor_ops=[m.actor_ops(i, envs[i]) for i in range(n_actors)])
    return m, tr, events


@pytest.mark.parametrize("kw", [dict(max_opts=30, warmup_period=40, sync_interval=7, record_agent_info_interval=4, record_compute_cost_interval=10),
                                dict(max_opts=12, warmup_period=5, sync_interval=1, record_agent_info_interval=0, record_compute_cost_interval=0),
                                dict(max_opts=20, warmup_period=64, sync_interval=20, record_agent_info_interval=5, record_compute_cost_interval=0)])
def test_loop_follows_the_reference_rules(kw):
    cfg = AsyncTrainerConfig(warmup_sleep_ms=1, **kw)
    man = ActorManagerConfig(n_buffer=8)
    m, tr, events = run(cfg, man, n_actors=3, opt_delay=0.0005)
    st = tr.stat
    # ---- learner (async_trainer/base.rs:299-388)
    assert m.log[0] == ("train", 1) and m.log[1] == ("publish", 0)              # agent.train(); "Send model info first"
    opts = [(k, e) for k, e in enumerate(m.log) if e[0] in ("opt", "opt_rec")]
    assert len(opts) == kw["max_opts"] == st.opt_steps
    assert all(e[1] >= kw["warmup_period"] for _, e in opts)                    # no opt before buffer.len() >= warmup_period
    rec = [j + 1 for j, (_, e) in enumerate(opts) if e[0] == "opt_rec"]
    k = kw["record_agent_info_interval"]
    assert rec == ([o for o in range(1, kw["max_opts"] + 1) if o % k == 0] if k else [])
    pubs = [e[1] for e in m.log if e[0] == "publish"]
    want = [0] + [o for o in range(1, kw["max_opts"] + 1) if o % kw["sync_interval"] == 0] + [kw["max_opts"]]   # + the final sync (:372)
    nonzero = [p for p in pubs if p > 0]
    assert nonzero == want[1:]
    assert all(p == 0 for p in pubs[:len(pubs) - len(nonzero)])                 # opt_steps % sync_interval == 0 also holds at 0 (warm-up loops)
    assert all(e == ("push", 8) for e in m.log if e[0] == "push")               # ReplayBufferProxy: messages of exactly n_buffer items
    assert st.samples_total == 8 * st.n_messages == sum(e[1] for e in m.log if e[0] == "push")
    assert st.n_records == len(rec) and st.n_syncs == len(pubs)
    assert abs(st.opt_per_sec - kw["max_opts"] / st.duration) < 1e-3 * st.opt_per_sec
    assert abs(st.samples_per_sec - st.samples_total / st.duration) < 1e-3 * max(st.samples_per_sec, 1)
    cost = [e for e in events if e[3] == "cost"]
    c = kw["record_compute_cost_interval"]
    if c:
        assert [e[2] for e in cost if e[2] > 0][:kw["max_opts"] // c] == [o for o in range(c, kw["max_opts"] + 1, c)][:len(cost)]
    # observer: push events carry the actor id and the running sample count
    push_ev = [e for e in events if e[3] == "push"]
    assert [e[1] for e in push_ev] == [8 * (j + 1) for j in range(len(push_ev))] and {e[0] for e in push_ev} <= {0, 1, 2}
    # ---- actors (actor/base.rs:120-178)
    for i in range(3):
        vs = m.actor_samples[i]
        assert len(vs) - tr.actor_stats[i].env_steps in (0, 1)                  # a sample whose env step was cut by the stop flag
        assert vs and vs[0] == 0                                                # sync_model_first before the first sample
        assert all(b >= a for a, b in zip(vs, vs[1:]))                          # only newer models are adopted
        ev = [e for e in events if e[3] == "actor_sync" and e[0] == i]
        assert ev[0][2] == 0 and [e[2] for e in ev] == sorted(set(e[2] for e in ev))
        assert tr.actor_stats[i].n_syncs == len(ev)
    # ---- transitions (SimpleStepProcessor through the proxy): per actor the obs chain continues, restarting after a done step
    chains = {1: [], 2: [], 3: []}
    for o, x, t in m.pushed:
        chains[int(o[0]) // 1000000].append((o, x, t))
    for ident, msgs in chains.items():
        o = np.concatenate([a for a, _, _ in msgs]); x = np.concatenate([b for _, b, _ in msgs]); t = np.concatenate([c for _, _, c in msgs])
        assert (o // 1000000 == ident).all()
        assert (o[1:] == np.where(t[:-1] == 1, x[:-1] + 1, x[:-1])).all()      # MockEnv: init_obs follows the terminal observation


def test_full_channel_is_an_error_like_send_msg_for_push():
    """ReplayBufferProxy::push uses try_send on a bounded channel; a full channel is BorderAsyncTrainerError::SendMsgForPush
    (the reference's actor thread panics on it).  Here: the run stops and bdr_async_train returns the error."""
    from border_amd import BdrError
    cfg = AsyncTrainerConfig(max_opts=10 ** 6, warmup_period=10 ** 9, sync_interval=10, warmup_sleep_ms=1)
    with pytest.raises(BdrError) as e:
        run(cfg, ActorManagerConfig(n_buffer=1, channel_capacity=1), n_actors=2, push_delay=0.05)
    assert "SendMsgForPush" in str(e.value)


def test_callback_errors_stop_every_thread():
    from border_amd import BdrError

    class BadEnv(MockEnv):
        def step_with_reset(self, act):
            if self.t > 20:
                raise RuntimeError("boom")
            return super().step_with_reset(act)
    m = Mock(2)
    envs = [MockEnv(1), BadEnv(2)]
    tr = AsyncTrainer(AsyncTrainerConfig(max_opts=10 ** 6, warmup_period=10 ** 9, sync_interval=5, warmup_sleep_ms=1), ActorManagerConfig(n_buffer=4))
    with pytest.raises(BdrError) as e:
        tr.train(None, None, [None, None], envs, (4,), np.float32, learner_ops=m.learner_ops(), actor_ops=[m.actor_ops(i, envs[i]) for i in range(2)])
    assert "actor env 1" in str(e.value)


# ---- world size 2 over gloo: the same compiled loop on every rank, learners averaged at every sync point --------------------
def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = Mock(2, opt_delay=0.0002)
    inc = float(rank + 1)
    # the learner's "parameters": opt adds rank+1; the exchange averages over ranks (what ParamExchange.average does for agents)
    base_ops = m.learner_ops()

    def opt(_a, _b):
        m.params += inc
        m.log.append(("opt", m.len)); return 0
    fn = _lib.OPT_FN(opt)
    m.keep.append(fn)
    base_ops.t.agent_opt = fn
    exchanged = []

    def exchange(opt_steps):
        t = torch.tensor([m.params], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        m.params = float(t[0]) / world
        exchanged.append(opt_steps)
    envs = [MockEnv(10 * rank + 1), MockEnv(10 * rank + 2)]
    tr = AsyncTrainer(AsyncTrainerConfig(max_opts=23, warmup_period=16, sync_interval=5, record_agent_info_interval=0,
                                         record_compute_cost_interval=0, warmup_sleep_ms=1), ActorManagerConfig(n_buffer=4))
    tr.train(None, None, [None, None], envs, (4,), np.float32, exchange=exchange, learner_ops=base_ops,
             actor_ops=[m.actor_ops(i, envs[i]) for i in range(2)])
    out.put((rank, m.params, exchanged, tr.stat.opt_steps, [v for v in m.actor_samples[0][-1:]]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_learners_are_averaged_at_every_sync_point():
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
    # sequential restatement: both ranks step 23 times (+1 / +2), averaging after steps 5, 10, 15, 20 and at the end
    p = [0.0, 0.0]
    for step in range(1, 24):
        p = [p[0] + 1.0, p[1] + 2.0]
        if step % 5 == 0:
            p = [sum(p) / 2] * 2
    p = [sum(p) / 2] * 2
    for rank, params, exchanged, opt_steps, _ in res:
        assert opt_steps == 23
        assert [e for e in exchanged if e > 0] == [5, 10, 15, 20, 23] and all(e == 0 for e in exchanged[:len(exchanged) - 5])
        assert abs(params - p[rank]) < 1e-9, (rank, params, p)
    assert res[0][1] == res[1][1]


# ---- a rank that fails between two sync points must not leave its peer in the next collective -------------------------------
def _worker_fail(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from border_amd._lib import BdrError
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = Mock(2, opt_delay=0.0002)
    base_ops = m.learner_ops()
    n = [0]

    def opt(_a, _b):
        n[0] += 1
        if rank == 1 and n[0] == 12:      # between the sync points of opt steps 10 and 15
            return 1                       # BDR_ERR_INVALID from Agent::opt
        m.params += 1.0
        return 0
    fn = _lib.OPT_FN(opt)
    m.keep.append(fn)
    base_ops.t.agent_opt = fn
    exchanged, agreed = [], []

    def exchange(opt_steps):
        t = torch.tensor([m.params], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        m.params = float(t[0]) / world
        exchanged.append(opt_steps)

    def agree(local_ok):
        t = torch.tensor([1 if local_ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        agreed.append((bool(local_ok), bool(int(t[0]))))
        return bool(int(t[0]))
    envs = [MockEnv(10 * rank + 1), MockEnv(10 * rank + 2)]
    tr = AsyncTrainer(AsyncTrainerConfig(max_opts=40, warmup_period=16, sync_interval=5, record_agent_info_interval=0,
                                         record_compute_cost_interval=0, warmup_sleep_ms=1), ActorManagerConfig(n_buffer=4))
    err = None
    try:
        tr.train(None, None, [None, None], envs, (4,), np.float32, exchange=exchange, agree=agree, learner_ops=base_ops,
                 actor_ops=[m.actor_ops(i, envs[i]) for i in range(2)])
    except BdrError as e:
        err = str(e)
    out.put((rank, err, exchanged, agreed))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_a_failed_rank_stops_its_peer_at_the_next_sync_point():
    """`exchange` is a collective inside sync(); without an agreement a rank whose learner fails returns while its peer
    blocks forever in the next all-reduce.  With bdr_learner_ops::agree (MIN of an ok flag before every collective) rank 1 fails
    at its 12th opt and says so once; rank 0 reaches the sync point of opt step 15, learns it and stops with BDR_ERR_COMM - both
    ranks return, neither enters the exchange of step 15."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fail, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=420) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, err0, ex0, ag0), (_, err1, ex1, ag1) = res
    assert err1 is not None and err0 is not None and "another rank's learner failed" in err0, (err0, err1)
    assert ex0 == ex1 == [0, 5, 10]                      # the first sync, steps 5 and 10; nobody entered the exchange of 15
    assert ag0 == [(True, True)] * 3 + [(True, False)]
    assert ag1 == [(True, True)] * 3 + [(False, False)]
