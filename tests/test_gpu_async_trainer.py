"""a16 on the GPU: the compiled async loops (csrc/async_trainer.hip) with real agents - one learner (DQN, Mlp) over its HBM
replay shard, two actors with their own agents, exploration streams and environments, the device-resident model mailbox.

The run is concurrent (three host threads, three HIP streams), so WHICH interleaving happens is up to the machine; the
observer records it (pushes with their actor, opt steps, syncs, and the env step at which every actor adopted which model).
The test then replays exactly that interleaving SEQUENTIALLY on the CPU oracle - oracle replay ring + oracle DQN update for
the learner, oracle forward + oracle explorer for every actor, each on the parameter version the trace says it had - and
requires: every ring row bit-identical (observations AND the actors' actions), the learner's index stream bit-identical, the
learner's parameters within 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def mlp_agent(B, seed=0, **kw):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2), opt_config=B.OptimizerConfig.Adam(1e-3)),
                      device=0, batch_size=16, critic_loss="Mse", tau=0.5, soft_update_interval=3, param_seed=seed, **kw)
    return B.Dqn.build(cfg)


def test_model_mailbox_publish_and_conditional_sync(B):
    """SyncModel through the device mailbox: sync_model_first is unconditional, later syncs only adopt a newer n_opts
    (actor/base.rs:98-118); the copy carries the learner's parameters exactly (target net and optimizer state stay local)."""
    learner, actor = mlp_agent(B, 1), mlp_agent(B, 2)
    box = B.ModelMailbox(learner, n_readers=1)
    with pytest.raises(B.BdrError):
        box.sync(actor, 0, 0, first=True)               # nothing published yet
    p1 = learner.get_params("qnet")
    tgt = actor.get_params("qnet_tgt").copy()
    box.publish(learner, 0)
    v, up = box.sync(actor, 0, 0, first=True)
    assert (v, up) == (0, True) and (actor.get_params("qnet") == p1).all() and (actor.get_params("qnet_tgt") == tgt).all()
    assert box.sync(actor, 0, 0) == (0, False)          # not newer
    p2 = (p1 * np.float32(0.5)).astype(np.float32)
    learner.set_params(p2, "qnet")
    box.publish(learner, 7)
    assert box.sync(actor, 0, 7) == (7, False)
    v, up = box.sync(actor, 0, 3)
    assert (v, up) == (7, True) and (actor.get_params("qnet") == p2).all()
    # publish overwrites the snapshot only after the reader's copy (events): back-to-back publish / sync pairs stay consistent
    for k in range(8, 40):
        learner.set_params((p1 * np.float32(k)).astype(np.float32), "qnet")
        box.publish(learner, k)
        assert box.sync(actor, 0, k - 1) == (k, True)
    assert (actor.get_params("qnet") == (p1 * np.float32(39)).astype(np.float32)).all()
    box.close(); learner.close(); actor.close()


@pytest.mark.parametrize("sync_interval,n_buffer", [(4, 8), (1, 5)])
def test_async_run_equals_its_sequential_replay_on_the_oracle(B, sync_interval, n_buffer):
    from oracle import oracle as O
    from oracle import torch_ref as T
    cap, Bsz, n_act, max_opts, warm = 200, 16, 2, 40, 64
    shapes = T.mlp_shapes(4, [64, 64], 2)
    p0 = T.init_params(shapes, 77)
    learner = mlp_agent(B)
    learner.set_params(p0, "qnet"); learner.set_params(p0, "qnet_tgt")
    actors = [mlp_agent(B, seed=10 + i) for i in range(n_act)]       # own (different) initial parameters: replaced by the first sync
    for i, a in enumerate(actors):
        a.set_explorer(B.EpsilonGreedy(final_step=60), seed=100 + i)
    envs = [B.SyntheticEnv((4,), np.float32, seed=i, p_term=0.1) for i in range(n_act)]
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4,), np.float32)
    events = []
    tr = B.AsyncTrainer(B.AsyncTrainerConfig(max_opts=max_opts, warmup_period=warm, sync_interval=sync_interval, record_agent_info_interval=0,
                                             record_compute_cost_interval=0, warmup_sleep_ms=5), B.ActorManagerConfig(n_buffer=n_buffer))
    st = tr.train(learner, rb, actors, envs, (4,), np.float32, on_event=lambda *e: events.append(e))
    assert st.opt_steps == max_opts == learner.n_opts and st.samples_total == n_buffer * st.n_messages and st.opt_per_sec > 0
    assert all(s.env_steps > 0 and s.n_syncs >= 1 for s in tr.actor_stats)

    # ---- sequential replay of the recorded interleaving on the oracle
    net = O.mlp_cfg(4, [64, 64], 2)
    ref = O.DqnOracle(net, p0, lr=1e-3, critic_loss="Mse", tau=0.5, soft_update_interval=3)
    oring = O.Replay(cap, 42, 16, 8)
    params_of = {}                                   # model version -> learner parameters at that sync

    class ActorSim:
        def __init__(self, i):
            self.env = B.SyntheticEnv((4,), np.float32, seed=i, p_term=0.1)
            self.explorer = O.Explorer("eps_greedy", final_step=60, seed=100 + i)
            self.syncs = [(e[1], e[2]) for e in events if e[3] == "actor_sync" and e[0] == i]    # (env step, version), in order
            self.env_steps, self.version, self.prev = 0, None, None

        def step(self):
            while self.syncs and self.syncs[0][0] == self.env_steps:       # "Check model update and synchronize" precedes the step
                self.version = self.syncs.pop(0)[1]
            if self.prev is None:
                self.prev = self.env.reset(None)
            q = O.net_forward(net, params_of[self.version], self.prev)
            act, _, _ = self.explorer.sample(q, train=True)
            s = self.env.step_with_reset(act)
            tr_ = (self.prev.copy(), act.copy(), s.obs.copy(), s.reward.copy(), s.is_terminated.copy(), s.is_truncated.copy())
            self.prev = s.init_obs.copy() if s.is_done() else s.obs.copy()
            self.env_steps += 1
            return tr_

    sims = [ActorSim(i) for i in range(n_act)]
    n_opt = 0
    for actor, a, b, ev, _ in events:
        if ev == "sync":
            params_of[b] = ref.q.copy()
            assert b == n_opt
        elif ev == "push":
            rows = [sims[actor].step() for _ in range(n_buffer)]
            oring.push(np.concatenate([r[0] for r in rows]), np.concatenate([r[1] for r in rows]).reshape(-1, 1), np.concatenate([r[2] for r in rows]),
                       np.concatenate([r[3] for r in rows]), np.concatenate([r[4] for r in rows]), np.concatenate([r[5] for r in rows]))
        elif ev in ("opt", "opt_record"):
            bt = oring.batch(Bsz)
            ref.update(bt["obs"].view(np.float32).reshape(Bsz, 4), bt["act"].view(np.int64).ravel(), bt["next_obs"].view(np.float32).reshape(Bsz, 4),
                       bt["reward"], bt["is_terminated"])
            n_opt += 1
    assert n_opt == max_opts and len(oring) == len(rb) and oring.head == rb.head
    # ring rows: bit-identical, including the actions the actors chose with the model versions they had
    e = oring.batch(1)
    assert rb.sample_indices(1).tolist() == e["ixs"].tolist()
    # index stream afterwards: the same draws were consumed on both sides
    assert rb.sample_indices(Bsz).tolist() == oring.batch(Bsz)["ixs"].tolist()
    # every stored row, through batches that cover the ring: fields bit-identical at identical indices
    for _ in range(40):
        g, w = rb.batch(Bsz), oring.batch(Bsz)
        assert (g.ix_sample == w["ixs"]).all()
        assert (g.obs.view(np.uint8).reshape(Bsz, -1) == w["obs"]).all() and (g.next_obs.view(np.uint8).reshape(Bsz, -1) == w["next_obs"]).all()
        assert (g.act.view(np.uint8).reshape(Bsz, -1) == w["act"]).all(), "an actor acted on a different model than the trace says"
        assert (g.reward == w["reward"]).all() and (g.is_terminated == w["is_terminated"]).all()
    assert rel(learner.get_params("qnet"), ref.q) < 1e-4, rel(learner.get_params("qnet"), ref.q)
    assert rel(learner.get_params("qnet_tgt"), ref.q_tgt) < 1e-4
    # the actors hold the last model the trace says they adopted
    for i, ag in enumerate(actors):
        last = [e_[2] for e_ in events if e_[3] == "actor_sync" and e_[0] == i][-1]
        assert rel(ag.get_params("qnet"), params_of[last]) < 1e-4
    for h in actors + [learner]:
        h.close()
    rb.close()


def test_async_sac_learner_and_actors(B):
    """a16 x a14: the compiled async loops with SAC handles - continuous f32 actions through the generic act rows
    (bdr_actor_ops_default dispatches Policy::sample by agent kind), SyncModel ships only `pi` (sac/base.rs:377-386) through the
    device mailbox.  Counters and interleaving rules as for DQN; the actors' pushed actions are valid tanh outputs, every actor
    adopts the learner's actor network, critics stay the actors' own, and the learner's losses are finite."""
    od, ad, n_act, max_opts, warm = 5, 2, 2, 30, 96
    def sac(seed):
        return B.Sac.build(B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=32,
                                       ent_coef_mode=("Auto", -2.0, 3e-4), device=0, seed=seed))
    learner = sac(1)
    actors = [sac(10 + i) for i in range(n_act)]
    q_before = [a.get_params("qnet_0").copy() for a in actors]
    envs = [B.SyntheticEnv((od,), np.float32, seed=i, p_term=0.1) for i in range(n_act)]
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=500, seed=42), (od,), np.float32, (ad,), np.float32)
    events = []
    tr = B.AsyncTrainer(B.AsyncTrainerConfig(max_opts=max_opts, warmup_period=warm, sync_interval=5, record_agent_info_interval=10,
                                             record_compute_cost_interval=0, warmup_sleep_ms=5), B.ActorManagerConfig(n_buffer=16))
    st = tr.train(learner, rb, actors, envs, (od,), np.float32, act_row_bytes=ad * 4, act_dtype=np.float32, on_event=lambda *e: events.append(e))
    assert st.opt_steps == max_opts and learner.n_opts == max_opts
    assert st.samples_total >= warm and st.samples_total % 16 == 0 and len(rb) == min(st.samples_total, 500)
    recs = [e for e in events if e[3] == "opt_record"]
    assert len(recs) == max_opts // 10
    for e in recs:
        assert all(np.isfinite(v) for v in e[4]), e
    # what the actors pushed: tanh-squashed actions; the ring rows came through the f32 act path
    b = rb.batch(64)
    assert b.act.dtype == np.float32 and b.act.shape == (64, ad) and (np.abs(b.act) <= 1.0).all() and np.abs(b.act).max() > 0
    # SyncModel: pi only.  Every actor ends on a published actor network (the last publish is the learner's final pi unless
    # the actor stopped before adopting it: it then holds an earlier version - compare with the sync events), critics untouched
    pi_final = learner.get_params("pi")
    synced = {e[0]: e[2] for e in events if e[3] == "actor_sync"}
    for i, a in enumerate(actors):
        assert (a.get_params("qnet_0") == q_before[i]).all()
        if synced.get(i) == max_opts:
            assert (a.get_params("pi") == pi_final).all()
    assert any(v > 0 for v in synced.values())
    for a in actors:
        a.close()
    learner.close(); rb.close()
