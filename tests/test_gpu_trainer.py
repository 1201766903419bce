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


@pytest.mark.parametrize("kind", ["mlp", "cnn"])
def test_native_driver_equals_the_python_loop(B, kind):
    """bdr_trainer_train (the compiled Trainer::train, csrc/trainer.hip) over the library's own agent and buffer handles, with
    the environment behind C callbacks, against the Python mirror driving the same objects: same seeds -> the same calls in
    the same order -> bit-identical parameters, buffer head and counters."""
    def build():
        if kind == "mlp":
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=50, seed=9), (4,), np.float32)
            q = B.MlpConfig(in_dim=4, units=(64, 64), out_dim=3)
            shape, dtype = (4,), np.float32
        else:
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=50, seed=9), (4, 1, 84, 84), np.uint8)
            q = B.AtariCnnConfig(n_stack=4, out_dim=6)
            shape, dtype = (4, 1, 84, 84), np.uint8
        cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=q, opt_config=B.OptimizerConfig.Adam(1e-3)), device=0, batch_size=8,
                          critic_loss="SmoothL1", tau=0.05, soft_update_interval=1, param_seed=4)
        a = B.Dqn.build(cfg)
        a.set_explorer(B.EpsilonGreedy(final_step=30), seed=3)
        env = B.SyntheticEnv(shape, dtype, seed=11, p_term=0.1)
        return rb, a, env, shape, dtype

    tc = dict(max_opts=12, opt_interval=2, warmup_period=9, record_agent_info_interval=5)
    rb1, a1, env1, shape, dtype = build()
    ev = []
    st = B.NativeTrainer(B.TrainerConfig(**tc)).train(env1, a1, rb1, shape, dtype, on_event=lambda e, o, k, sc: ev.append((e, o, k, sc)))
    a1.sync()
    rb2, a2, env2, _, _ = build()
    tr = B.Trainer(B.TrainerConfig(**tc))
    recs = []
    tr.train(env2, B.SimpleStepProcessor(), a2, rb2, on_step=lambda s, r, o: recs.append(r) if r is not None else None)
    a2.sync()
    assert st["opt_steps"] == tr.opt_steps == 12 and st["env_steps"] == tr.env_steps
    assert rb1.len() == rb2.len() and rb1.head == rb2.head
    assert (a1.get_params("qnet") == a2.get_params("qnet")).all() and (a1.get_params("qnet_tgt") == a2.get_params("qnet_tgt")).all()
    assert a1.n_opts == a2.n_opts == 12
    native_losses = [sc[0] for _, _, k, sc in ev if k == "opt_record"]
    assert len(native_losses) == len(recs) == 2 and native_losses == [np.float32(r["loss"]) for r in recs]
    for x in (a1, a2, rb1, rb2):
        x.close()


def test_native_driver_with_a_continuous_action_agent(B):
    """bdr_trainer_train with SAC handles: the default function table samples f32 action rows (bdr_sac_sample) and pushes them
    through the same generic act rows; the loop rules (warm-up, opt_interval, record interval, counters) are the DQN ones."""
    od, ad = 5, 2
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=200, seed=9), (od,), np.float32, (ad,), np.float32)
    a = B.Sac.build(B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=16,
                                ent_coef_mode=("Auto", -2.0, 3e-4), device=0, seed=3))
    env = B.SyntheticEnv((od,), np.float32, seed=11, p_term=0.1)
    ev = []
    st = B.NativeTrainer(B.TrainerConfig(max_opts=20, opt_interval=2, warmup_period=24, record_agent_info_interval=5)).train(
        env, a, rb, (od,), np.float32, act_row_bytes=ad * 4, act_dtype=np.float32, on_event=lambda e, o, k, sc: ev.append((e, o, k, sc)))
    a.sync()
    assert st["opt_steps"] == a.n_opts == 20 and st["env_steps"] == rb.len() and 24 + 2 * 19 <= st["env_steps"] <= 24 + 2 * 20
    recs = [sc for _, _, k, sc in ev if k == "opt_record"]
    assert len(recs) == 4 and all(np.isfinite(v) for sc in recs for v in sc)
    b = rb.batch(32)
    assert b.act.dtype == np.float32 and (np.abs(b.act) <= 1.0).all() and np.abs(b.act).max() > 0
    a.close(); rb.close()


def test_compiled_example_program_trains(B):
    """examples/train_dqn_synthetic.cpp: a complete training program in compiled code on nothing but include/border_amd.h
    (environment callbacks, bdr_trainer_train, Nature-CNN DQN, HBM ring) - the integration a Rust shim would perform."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "..", "examples", "train_dqn_synthetic")
    if not os.path.exists(exe):
        from border_amd import build
        build.build_examples()
    out = subprocess.run([exe, "120"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    m = re.search(r"done: env_steps (\d+) opt_steps (\d+) episodes (\d+) buffer_len (\d+) n_opts (\d+)", out.stdout)
    assert m, out.stdout[-500:]
    env_steps, opt_steps, episodes, buffer_len, n_opts = map(int, m.groups())
    assert opt_steps == n_opts == 120 and env_steps == 64 + 119 and buffer_len == env_steps     # warm-up 64, then one opt per step
    losses = [float(x) for x in re.findall(r"loss ([0-9.eE+-]+)", out.stdout)]
    assert len(losses) == 2 and all(np.isfinite(losses))


def test_compiled_online_loop_with_device_observations(B):
    """examples/online_loop_atari.cpp: the one-environment loop in compiled code with the observation resident in HBM
    (bdr_atari_prep -> obs_on_device environment -> bdr_trainer_train): counters of trainer.rs:267-327, both conventions."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "..", "examples", "online_loop_atari")
    if not os.path.exists(exe):
        from border_amd import build
        build.build_examples()
    for where in ("device", "host"):
        out = subprocess.run([exe, "200", "32", "0", where], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:] + out.stdout[-500:]
        m = re.search(r"done: env_steps (\d+) opt_steps (\d+) episodes (\d+) buffer_len (\d+) n_opts (\d+)", out.stdout)
        assert m, out.stdout[-500:]
        env_steps, opt_steps, episodes, buffer_len, n_opts = map(int, m.groups())
        assert opt_steps == n_opts == 200 and env_steps == 512 + 199 and buffer_len == env_steps
        assert ("observations on the " + where) in out.stdout and re.search(r"= [0-9.]+ it/s", out.stdout)
