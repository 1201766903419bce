"""Host logic of the online loop (no GPU): SimpleStepProcessor (generic_replay_buffer/step_proc.rs:62-137),
Sampler::sample_and_push (trainer/sampler.rs:99-144) and the gating of Trainer::train / train_step
(trainer.rs:197-228, 267-327) with stand-in agent / buffer objects."""
import numpy as np

from border_amd.trainer import Sampler, SimpleStepProcessor, Step, SyntheticEnv, Trainer, TrainerConfig


class FakeBuffer:
    def __init__(self):
        self.rows = []

    def push(self, obs, act, next_obs, reward, term, trunc):
        self.rows.append((obs.copy(), np.asarray(act).copy(), next_obs.copy(), float(reward[0]), int(term[0]), int(trunc[0])))


class FakeAgent:
    def __init__(self):
        self.calls, self.n_sample, self.is_training = [], 0, False

    def train(self):
        self.is_training = True

    def sample(self, obs):
        self.n_sample += 1
        return np.array([self.n_sample % 3], np.int64)

    def opt(self, buffer):
        self.calls.append(("opt", len(buffer.rows)))

    def opt_with_record(self, buffer):
        self.calls.append(("rec", len(buffer.rows)))
        return {"loss": 0.0}


class ScriptedEnv:
    """obs = [t]; terminal at t in `ends`; init_obs after a terminal = [100 + t]."""

    def __init__(self, ends):
        self.t, self.ends = 0, set(ends)

    def reset(self, is_done=None):
        return np.array([[-1.0]], np.float32)

    def step_with_reset(self, act):
        self.t += 1
        done = self.t in self.ends
        st = Step(np.asarray(act), np.array([[float(self.t)]], np.float32), np.array([1.0], np.float32),
                  np.array([1 if done else 0], np.int8), np.array([0], np.int8))
        if done:
            st.init_obs = np.array([[100.0 + self.t]], np.float32)
        return st


def test_step_processor_chains_observations_and_restarts_after_done():
    env, buf, agent = ScriptedEnv(ends=[3]), FakeBuffer(), FakeAgent()
    s = Sampler(env, SimpleStepProcessor())
    for _ in range(5):
        s.sample_and_push(agent, buf)
    obs = [r[0][0, 0] for r in buf.rows]
    nxt = [r[2][0, 0] for r in buf.rows]
    assert obs == [-1.0, 1.0, 2.0, 103.0, 4.0]      # after the terminal step the chain restarts from init_obs (:122-126)
    assert nxt == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert [r[4] for r in buf.rows] == [0, 0, 1, 0, 0]
    assert [int(r[1][0]) for r in buf.rows] == [1, 2, 0, 1, 2]


def test_trainer_train_gating():
    """env_steps < warmup_period or env_steps % opt_interval != 0 -> no opt; every record_agent_info_interval-th opt
    is opt_with_record; stop at opt_steps == max_opts (trainer.rs:197-228, 322-325)."""
    env, buf, agent = ScriptedEnv(ends=[]), FakeBuffer(), FakeAgent()
    t = Trainer(TrainerConfig(max_opts=6, opt_interval=2, warmup_period=5, record_agent_info_interval=3))
    t.train(env, SimpleStepProcessor(), agent, buf)
    assert agent.is_training and t.opt_steps == 6
    # opts happen at env_steps 6, 8, 10, 12, 14, 16 (the buffer holds env_steps rows at that point)
    assert [n for _, n in agent.calls] == [6, 8, 10, 12, 14, 16]
    assert [k for k, _ in agent.calls] == ["opt", "opt", "rec", "opt", "opt", "rec"]
    assert t.env_steps == 16 and len(t.records) == 2 and [o for o, _ in t.records] == [3, 6]


def test_synthetic_env_is_seeded_and_typed():
    a, b = SyntheticEnv((4, 1, 84, 84), np.uint8, seed=3), SyntheticEnv((4, 1, 84, 84), np.uint8, seed=3)
    o1, o2 = a.reset(), b.reset()
    assert o1.dtype == np.uint8 and o1.shape == (1, 4, 1, 84, 84) and (o1 == o2).all()
    s1, s2 = a.step_with_reset(np.array([1])), b.step_with_reset(np.array([1]))
    assert (s1.obs == s2.obs).all() and s1.reward[0] == s2.reward[0]
    f = SyntheticEnv((4,), np.float32, seed=1, p_term=1.0)
    st = f.step_with_reset(np.array([0]))
    assert st.is_done() and st.init_obs is not None and st.obs.dtype == np.float32
