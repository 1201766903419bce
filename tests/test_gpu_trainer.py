"""The online loop end to end on the device path (Trainer::train, trainer.rs:267-327): Policy::sample -> env step ->
SimpleStepProcessor -> push -> gated opt, replayed on the CPU restatements (oracle ring + ATen update) with the same
transitions.  What this pins beyond the unit tests: pushes and gathers interleave on different streams, and every
batch must see exactly the rows the reference's sequential loop would see."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


@pytest.mark.parametrize("per", [False, True])
def test_online_loop_matches_sequential_restatement(B, per):
    from oracle import oracle as O
    from oracle import torch_ref as T
    cap, Bsz = 64, 8                       # small ring: it wraps during the run
    shapes = T.mlp_shapes(4, [64, 64], 3)
    p0 = T.init_params(shapes, 51)
    percfg = B.PerConfig(alpha=1.0, normalize="All", n_opts_final=20) if per else None
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=5, per_config=percfg), (4,), np.float32)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=3),
                                                    opt_config=B.OptimizerConfig.Adam(1e-3)),
                      device=0, batch_size=Bsz, critic_loss="SmoothL1", tau=0.05, soft_update_interval=1)
    a = B.Dqn.build(cfg)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    a.set_explorer(B.EpsilonGreedy(final_step=40), seed=2)
    t = T.TorchDqn("mlp", shapes, p0, lr=1e-3, critic_loss="SmoothL1", tau=0.05, soft_update_interval=1)
    oref = O.Replay(cap, 5, 16, 8)
    pref = O.PerReplay(cap, 5, alpha=1.0, normalize="All", n_opts_final=20) if per else None
    rows = {k: np.zeros((cap,) + s, d) for k, s, d in [("obs", (4,), np.float32), ("nobs", (4,), np.float32), ("act", (), np.int64),
                                                         ("rew", (), np.float32), ("term", (), np.int8)]}
    state = {"i": 0, "prev": None, "n": 0}
    losses = []

    def on_step(step, rec, is_opt):
        # the transition the reference's sequential loop would have pushed before this opt
        obs = state["prev"]
        i = state["i"]
        rows["obs"][i], rows["nobs"][i], rows["act"][i] = obs[0], step.obs[0], int(step.act[0])
        rows["rew"][i], rows["term"][i] = step.reward[0], step.is_terminated[0]
        oref.push(obs, np.asarray(step.act, np.int64).reshape(1, 1), step.obs, step.reward, step.is_terminated, step.is_truncated)
        if per:
            pref.push(1)
        state["i"] = (i + 1) % cap
        state["prev"] = step.init_obs if step.is_done() else step.obs
        if not is_opt:
            return
        if per:
            ixs, ws = pref.batch(Bsz)
        else:
            ixs, ws = oref.batch(Bsz)["ixs"].astype(np.int64), None
        r = t.update(rows["obs"][ixs], rows["act"][ixs], rows["nobs"][ixs], rows["rew"][ixs], rows["term"][ixs], weight=ws)
        if per:
            pref.update_priority(ixs, r["td_abs"])
        assert abs(rec["loss"] - r["loss"]) <= 3e-4 * abs(r["loss"]) + 1e-7, (state["n"], rec["loss"], r["loss"])
        losses.append(rec["loss"])
        state["n"] += 1

    env = B.SyntheticEnv((4,), np.float32, seed=9, p_term=0.15)
    # the sampler resets the env first: mirror that observation as the first `prev`
    probe = B.SyntheticEnv((4,), np.float32, seed=9, p_term=0.15)
    state["prev"] = probe.reset()
    tr = B.Trainer(B.TrainerConfig(max_opts=45, opt_interval=2, warmup_period=10, record_agent_info_interval=1))
    tr.train(env, B.SimpleStepProcessor(), a, rb, on_step=on_step)
    assert tr.opt_steps == 45 and len(losses) == 45 and tr.env_steps == 10 + 2 * 44      # opts at env steps 10, 12, ..., 98
    assert len(rb) == cap and rb.head == tr.env_steps % cap
    assert np.abs(a.get_params("qnet").astype(np.float64) - t.params()).max() < 2e-4
    a.close(); rb.close()
