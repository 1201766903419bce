"""The compiled Trainer loops (csrc/trainer.hip) against the rules of border-core/src/trainer.rs:197-228, 267-384 and
trainer/sampler.rs:99-144, driven with mock agent / buffer / environment callbacks - no GPU involved - and against the
Python mirror (border_amd/trainer.py), which restates the same rules."""
import ctypes as C

import numpy as np
import pytest

from border_amd import _lib
from border_amd.trainer import NativeTrainer, SimpleStepProcessor, Step, Trainer, TrainerConfig


class MockEnv:
    """Deterministic env: obs = [t, t, t, t] f32 of the step counter; done every `period` steps."""
    def __init__(self, period=5):
        self.t, self.period = 0, period

    def _obs(self):
        self.t += 1
        return np.full((1, 4), self.t, np.float32)

    def reset(self, _=None):
        return self._obs()

    def step_with_reset(self, act):
        obs = self._obs()
        done = (self.t % self.period) == 0
        st = Step(np.asarray(act), obs, np.array([0.5 * self.t], np.float32), np.array([1 if done else 0], np.int8),
                  np.array([0], np.int8))
        if done:
            st.init_obs = self.reset()
        return st


class MockAgentBuffer:
    """Records every call; sample returns n_calls as the action."""
    def __init__(self):
        self.log, self.pushed, self.n_sample = [], [], 0

    def ops(self):
        def set_train(_a, on):
            self.log.append(("train", on)); return 0

        def sample(_a, n, obs, act_out):
            o = np.frombuffer((C.c_char * 16).from_address(obs), np.float32).copy()
            self.n_sample += 1
            self.log.append(("sample", float(o[0])))
            C.cast(act_out, C.POINTER(C.c_int64))[0] = self.n_sample
            return 0

        def opt(_a, _b):
            self.log.append(("opt",)); return 0

        def opt_rec(_a, _b, out, cap, n):
            self.log.append(("opt_rec",)); out[0] = 1.25; out[1] = -2.0; n[0] = 2; return 0

        def push(_b, n, obs, act, nobs, rew, term, trunc):
            o = np.frombuffer((C.c_char * 16).from_address(obs), np.float32)[0]
            x = np.frombuffer((C.c_char * 16).from_address(nobs), np.float32)[0]
            a = C.cast(act, C.POINTER(C.c_int64))[0]
            self.pushed.append((float(o), int(a), float(x), float(rew[0]), int(term[0]), int(trunc[0])))
            self.log.append(("push",)); return 0

        self._keep = (_lib.SET_TRAIN_FN(set_train), _lib.SAMPLE_FN(sample), _lib.OPT_FN(opt), _lib.OPT_REC_FN(opt_rec), _lib.PUSH_FN(push))
        return _lib.TrainerOps(None, None, *self._keep)

    # the same object seen through the Python mirror's interface
    def train(self): self.log.append(("train", 1))
    def sample(self, obs):
        self.n_sample += 1; self.log.append(("sample", float(obs[0, 0]))); return np.array([self.n_sample], np.int64)
    def opt(self, _b): self.log.append(("opt",))
    def opt_with_record(self, _b): self.log.append(("opt_rec",)); return {"loss": 1.25}
    def push(self, obs, act, nobs, rew, term, trunc):
        self.pushed.append((float(obs[0, 0]), int(act[0]), float(nobs[0, 0]), float(rew[0]), int(term[0]), int(trunc[0])))
        self.log.append(("push",))


@pytest.mark.parametrize("cfg", [dict(max_opts=7, opt_interval=1, warmup_period=0, record_agent_info_interval=3),
                                 dict(max_opts=5, opt_interval=4, warmup_period=10, record_agent_info_interval=2),
                                 dict(max_opts=3, opt_interval=3, warmup_period=7, record_agent_info_interval=0)])
def test_online_loop_matches_the_reference_rules_and_the_python_mirror(cfg):
    # native
    m = MockAgentBuffer()
    events = []
    nt = NativeTrainer(TrainerConfig(**cfg))
    st = nt.train(MockEnv(), None, None, (4,), np.float32, on_event=lambda e, o, ev, sc: events.append((e, o, ev, sc)), ops=m.ops())
    # mirror
    m2 = MockAgentBuffer()
    tr = Trainer(TrainerConfig(**cfg))
    tr.train(MockEnv(), SimpleStepProcessor(), m2, m2)
    assert m.log == m2.log and m.pushed == m2.pushed
    assert st["opt_steps"] == tr.opt_steps == cfg["max_opts"] and st["env_steps"] == tr.env_steps
    # the rules themselves (trainer.rs:197-228)
    opt_at = [e for e, o, ev, _ in events if ev in ("opt", "opt_record")]
    first = max(cfg["warmup_period"], 1)
    first += (-first) % cfg["opt_interval"]
    assert opt_at == [first + k * cfg["opt_interval"] for k in range(cfg["max_opts"])]
    rec_at = [o for e, o, ev, _ in events if ev == "opt_record"]
    k = cfg["record_agent_info_interval"]
    assert rec_at == ([o for o in range(1, cfg["max_opts"] + 1) if o % k == 0] if k else [])
    assert all(sc == [1.25, -2.0] for _, _, ev, sc in events if ev == "opt_record")
    assert st["n_records"] == len(rec_at)
    # transitions (step_proc.rs:103-137): obs chain continues from next_obs, restarts from init_obs after a done step
    for (o, a, x, r, t, _), nxt in zip(m.pushed, m.pushed[1:]):
        assert nxt[0] == (x + 1 if t else x)          # MockEnv: init_obs is the observation after the terminal one
    assert [p[1] for p in m.pushed] == list(range(1, len(m.pushed) + 1))      # one Policy::sample per env step, in order
    assert st["n_episodes"] == sum(p[4] for p in m.pushed)
    assert m.log[0] == ("train", 1)


def test_offline_loop_and_cost_records():
    m = MockAgentBuffer()
    events = []
    nt = NativeTrainer(TrainerConfig(max_opts=10, opt_interval=7, warmup_period=99, record_agent_info_interval=4, record_compute_cost_interval=5))
    st = nt.train_offline(None, None, on_event=lambda e, o, ev, sc: events.append((e, o, ev, sc)), ops=m.ops())
    # trainer.rs:345-346: warmup_period = 0 and opt_interval = 1 whatever the config says
    assert st["opt_steps"] == st["env_steps"] == 10
    assert [x[0] for x in m.log] == ["train"] + ["opt", "opt", "opt", "opt_rec"] * 2 + ["opt", "opt"]
    cost = [(o, sc) for _, o, ev, sc in events if ev == "cost"]
    assert [o for o, _ in cost] == [5, 10] and all(len(sc) == 2 and sc[0] >= 0 and sc[1] == -1.0 for _, sc in cost)


def test_argument_checks():
    from border_amd import BdrError
    m = MockAgentBuffer()
    with pytest.raises(BdrError):
        NativeTrainer(TrainerConfig(max_opts=0)).train_offline(None, None, ops=m.ops())
    with pytest.raises(BdrError):
        NativeTrainer(TrainerConfig(max_opts=1, opt_interval=0)).train_offline(None, None, ops=m.ops())


def test_callback_errors_stop_the_loop_and_propagate():
    """A non-zero status from any callback ends the loop at once and is what bdr_trainer_train returns (the reference's `?`)."""
    from border_amd import BdrError
    m = MockAgentBuffer()
    ops = m.ops()
    calls = {"n": 0}

    class FailingEnv(MockEnv):
        def step_with_reset(self, act):
            calls["n"] += 1
            if calls["n"] == 4:
                raise RuntimeError("boom")
            return super().step_with_reset(act)

    # exceptions inside a ctypes callback cannot cross the C frame: turn them into a status code like a Rust shim would
    env = FailingEnv()
    orig = env.step_with_reset

    def guarded(act):
        try:
            return orig(act)
        except RuntimeError:
            return None
    env.step_with_reset = guarded
    nt = NativeTrainer(TrainerConfig(max_opts=50))
    row = 16

    def reset(_c, obs_out):
        C.memmove(obs_out, env.reset(None).ctypes.data, row); return 0

    def step(_c, act, obs_out, reward, term, trunc, init_out):
        st = env.step_with_reset(np.zeros(1, np.int64))
        if st is None:
            return 42
        C.memmove(obs_out, st.obs.ctypes.data, row)
        reward[0], term[0], trunc[0] = float(st.reward[0]), int(st.is_terminated[0]), 0
        if st.is_done():
            C.memmove(init_out, st.init_obs.ctypes.data, row)
        return 0

    vt = _lib.EnvVtable(None, _lib.ENV_RESET_FN(reset), _lib.ENV_STEP_FN(step))
    c = nt._config(row, 8)
    st = _lib.TrainerStatsC()
    rc = _lib.lib().bdr_trainer_train(C.byref(c), C.byref(ops), C.byref(vt), _lib.OBSERVER_FN(lambda *a: None), None, C.byref(st))
    assert rc == 42
    assert [x[0] for x in m.log].count("push") == 3 and [x[0] for x in m.log].count("opt") == 3   # three full iterations, then the failing step
    with pytest.raises(BdrError):
        _lib.check(rc)
