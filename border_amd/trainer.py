"""Host-side mirror of the opt-step loops that call the hot path.

  Trainer.train_offline   border-core/src/trainer.rs:330-384 (pre-filled buffer, back-to-back
                          Agent::opt; warmup_period = 0, opt_interval = 1) and the gating / timing
                          of Trainer::train_step (:197-228): every record_agent_info_interval-th
                          step is opt_with_record, the timer wraps opt*() only.
  Trainer.train           trainer.rs:267-327: the online loop - Sampler::sample_and_push (trainer/sampler.rs:99-144:
                          Policy::sample on the device, env step, SimpleStepProcessor, push) then train_step.
  SimpleStepProcessor     generic_replay_buffer/step_proc.rs:62-137.
  SyntheticEnv            the benchmark's stand-in environment (envs stay in the caller).
  ParamExchange           the N>1 replacement of border-async-trainer's learner->actors model
                          channel (async_trainer/base.rs:268-272): one replica per GPU, local replay
                          shard, parameter averaging every sync_interval opt steps over RCCL.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class TrainerConfig:
    """border-core/src/trainer/config.rs:30-87 (fields used by the offline loop)."""
    max_opts: int = 0
    opt_interval: int = 1
    record_agent_info_interval: int = 0   # 0: never (the reference default is usize::MAX-like "0 => unset")
    record_compute_cost_interval: int = 0
    warmup_period: int = 0


@dataclass
class Step:
    """border-core/src/base/step.rs:68-92 (single-process env: every field has one row)."""
    act: np.ndarray
    obs: np.ndarray
    reward: np.ndarray          # f32 [1]
    is_terminated: np.ndarray   # i8 [1]
    is_truncated: np.ndarray    # i8 [1]
    init_obs: "np.ndarray | None" = None

    def is_done(self) -> bool:  # step.rs:136-138
        return bool(self.is_terminated[0] == 1 or self.is_truncated[0] == 1)


class SyntheticEnv:
    """Stand-in for `Env` (border-core/src/base/env.rs:45-181): i.i.d. observations from a seeded generator, rewards
    and episode ends independent of the action.  obs rows have the replay buffer's row type."""

    def __init__(self, obs_shape, obs_dtype, seed: int = 0, p_term: float = 0.02, p_trunc: float = 0.0):
        self.obs_shape, self.obs_dtype = tuple(obs_shape), np.dtype(obs_dtype)
        self.rng = np.random.default_rng(seed)
        self.p_term, self.p_trunc = p_term, p_trunc

    def _obs(self):
        if self.obs_dtype == np.uint8:
            return self.rng.integers(0, 256, (1,) + self.obs_shape, dtype=np.uint8)
        return self.rng.standard_normal((1,) + self.obs_shape).astype(self.obs_dtype)

    def reset(self, is_done=None):
        return self._obs()

    def step_with_reset(self, act) -> Step:     # env.rs:137-161: the step; on done, init_obs = reset()
        obs = self._obs()
        rew = np.array([self.rng.choice([-1.0, 0.0, 1.0], p=[0.05, 0.9, 0.05])], np.float32)
        term = np.array([1 if self.rng.random() < self.p_term else 0], np.int8)
        trunc = np.array([1 if self.rng.random() < self.p_trunc else 0], np.int8)
        st = Step(np.asarray(act), obs, rew, term, trunc)
        if st.is_done():
            st.init_obs = self.reset()
        return st


class SimpleStepProcessor:
    """generic_replay_buffer/step_proc.rs:62-137: (prev_obs, act, obs, ...) transitions; after a terminal step the
    next transition starts from init_obs."""

    def __init__(self):
        self.prev_obs = None

    def reset(self, init_obs):
        self.prev_obs = init_obs

    def process(self, step: Step):
        assert len(step.obs) == 1
        if self.prev_obs is None:
            raise RuntimeError("prev_obs is not set. Forgot to call reset()?")
        obs, self.prev_obs = self.prev_obs, step.obs
        if step.is_done():
            assert step.init_obs is not None, "Failed to unwrap init_obs"
            self.prev_obs = step.init_obs
        return obs, step.act, step.obs, step.reward, step.is_terminated, step.is_truncated


class Sampler:
    """trainer/sampler.rs:99-144."""

    def __init__(self, env, step_proc):
        self.env, self.step_processor, self.prev_obs = env, step_proc, None

    def sample_and_push(self, agent, buffer):
        if self.prev_obs is None:
            self.prev_obs = self.env.reset(None)
            self.step_processor.reset(self.prev_obs.copy())
        act = agent.sample(self.prev_obs)                       # Policy::sample on the device
        step = self.env.step_with_reset(act)
        is_done = step.is_done()
        self.prev_obs = step.init_obs.copy() if is_done else step.obs.copy()
        buffer.push(*self.step_processor.process(step))
        if is_done:
            self.step_processor.reset(self.prev_obs.copy())
        return step


class Trainer:
    def __init__(self, config: TrainerConfig):
        self.config = config
        self.env_steps = 0
        self.opt_steps = 0
        self.timer_for_opt_steps = 0.0
        self.opt_steps_counter = 0
        self.records = []

    def train_step(self, agent, buffer):
        """trainer.rs:197-228."""
        c = self.config
        if self.env_steps < c.warmup_period or self.env_steps % c.opt_interval != 0:
            return None, False
        t0 = time.perf_counter()
        if c.record_agent_info_interval and (self.opt_steps + 1) % c.record_agent_info_interval == 0:
            rec = agent.opt_with_record(buffer)
        else:
            agent.opt(buffer)
            rec = None
        self.opt_steps += 1
        self.timer_for_opt_steps += time.perf_counter() - t0
        self.opt_steps_counter += 1
        return rec, True

    def average_opt_time_ms(self):
        """trainer.rs:164-174 (host-side enqueue time unless the step synchronised)."""
        return 1000.0 * self.timer_for_opt_steps / max(1, self.opt_steps_counter)

    def train(self, env, step_proc, agent, buffer, on_step=None):
        """trainer.rs:267-327 without recorder / evaluator sinks: sample_and_push, then train_step, until max_opts.
        `on_step(step, record, is_opt)` observes every iteration (tests)."""
        sampler = Sampler(env, step_proc)
        agent.train()
        self.timer_for_samples = 0.0
        while True:
            t0 = time.perf_counter()
            step = sampler.sample_and_push(agent, buffer)
            self.timer_for_samples += time.perf_counter() - t0
            self.env_steps += 1
            rec, is_opt = self.train_step(agent, buffer)
            if rec is not None:
                self.records.append((self.opt_steps, rec))
            if on_step is not None:
                on_step(step, rec, is_opt)
            if self.opt_steps == self.config.max_opts:
                return

    def train_offline(self, agent, buffer, exchange=None):
        """trainer.rs:330-384 without recorder / evaluator sinks (out of scope: SURVEY.md 2.1 #4)."""
        self.config.warmup_period = 0
        self.config.opt_interval = 1
        agent.train()
        while True:
            self.env_steps += 1
            rec, is_opt = self.train_step(agent, buffer)
            if rec is not None:
                self.records.append((self.opt_steps, rec))
            if is_opt and exchange is not None:
                exchange.after_opt(agent, self.opt_steps)
            if self.opt_steps == self.config.max_opts:
                return


class NativeTrainer:
    """The compiled driver (csrc/trainer.hip: bdr_trainer_train / bdr_trainer_train_offline) - the loops above run in C++;
    Python only supplies the environment callbacks and, optionally, an observer."""

    EVENTS = {0: "skip", 1: "opt", 2: "opt_record", 3: "cost"}

    def __init__(self, config: TrainerConfig):
        self.config = config
        self.stats = None

    def _config(self, obs_row_bytes=0, act_row_bytes=0):
        c = _lib.TrainerConfigC()
        _lib.lib().bdr_trainer_config_default(C.byref(c))
        k = self.config
        c.max_opts, c.opt_interval, c.warmup_period = k.max_opts, k.opt_interval, k.warmup_period
        c.record_agent_info_interval, c.record_compute_cost_interval = k.record_agent_info_interval, k.record_compute_cost_interval
        c.obs_row_bytes, c.act_row_bytes = obs_row_bytes, act_row_bytes
        return c

    @staticmethod
    def _observer(on_event):
        def cb(_ctx, env_steps, opt_steps, event, scalars, n):
            if on_event is not None:
                on_event(env_steps, opt_steps, NativeTrainer.EVENTS[event], [scalars[i] for i in range(n)])
        return _lib.OBSERVER_FN(cb)

    def _finish(self, st):
        self.stats = {k: getattr(st, k) for k, _ in _lib.TrainerStatsC._fields_}
        return self.stats

    def train_offline(self, agent, buffer, on_event=None, ops=None):
        if ops is None:
            ops = _lib.TrainerOps()
            _lib.lib().bdr_trainer_ops_default(C.byref(ops), agent.handle, buffer.handle)
        c, st, obs = self._config(), _lib.TrainerStatsC(), self._observer(on_event)
        _lib.check(_lib.lib().bdr_trainer_train_offline(C.byref(c), C.byref(ops), obs, None, C.byref(st)))
        return self._finish(st)

    def train(self, env, agent, buffer, obs_shape, obs_dtype, act_row_bytes=8, on_event=None, ops=None, act_dtype=np.int64):
        """`env` has reset(None) -> obs[1, ...] and step_with_reset(act) -> Step (as SyntheticEnv).  Continuous-action agents
        (SAC): act_row_bytes = 4 * act_dim, act_dtype = np.float32."""
        obs_dtype = np.dtype(obs_dtype)
        row = int(np.prod(obs_shape)) * obs_dtype.itemsize

        def write(ptr, arr):
            C.memmove(ptr, np.ascontiguousarray(arr, obs_dtype).ctypes.data, row)

        def reset(_ctx, obs_out):
            write(obs_out, env.reset(None))
            return 0

        def step(_ctx, act, obs_out, reward, term, trunc, init_out):
            a = np.frombuffer((C.c_char * act_row_bytes).from_address(act), act_dtype).copy()
            st = env.step_with_reset(a)
            write(obs_out, st.obs)
            reward[0], term[0], trunc[0] = float(st.reward[0]), int(st.is_terminated[0]), int(st.is_truncated[0])
            if st.is_done():
                write(init_out, st.init_obs)
            return 0

        vt = _lib.EnvVtable(None, _lib.ENV_RESET_FN(reset), _lib.ENV_STEP_FN(step))
        if getattr(env, "device_obs", False):   # device-resident observations: the callbacks get device buffers (see async_trainer.env_vtable)
            from .async_trainer import env_vtable
            self._keep = []
            vt = env_vtable(env, obs_shape, obs_dtype, act_row_bytes, act_dtype, keep=self._keep)
        if ops is None:
            ops = _lib.TrainerOps()
            _lib.lib().bdr_trainer_ops_default(C.byref(ops), agent.handle, buffer.handle)
        c, st, obs = self._config(row, act_row_bytes), _lib.TrainerStatsC(), self._observer(on_event)
        _lib.check(_lib.lib().bdr_trainer_train(C.byref(c), C.byref(ops), C.byref(vt), obs, None, C.byref(st)))
        return self._finish(st)


class ParamExchange:
    """Parameter averaging across one-replica-per-GPU ranks over the library's own communicator (csrc/comm.hip): ncclAllReduce on the
    flat f32 arena + 1/N scale, enqueued on the agent's stream (or, per segment, on its communication queue); no host sync.  The
    unique id is created on rank 0 and handed to the other ranks by `bcast_bytes` (a callable(bytes|None) -> bytes, e.g. a
    torch.distributed broadcast on the control plane).  There is no other data plane here: rounds 1-5 carried a host-vector
    "torch" backend for the CPU tests; those now subclass this class (tests/test_param_exchange_gloo.py) and the N ranks-on-one-GPU
    flow tests load the communicator's host-transport test build instead (libborder_amd_hostcomm.so, build.build_hostcomm_library)."""

    def __init__(self, world_size: int, rank: int, sync_interval: int, device: int = 0, bcast_bytes=None, which=("qnet",)):
        self.world_size, self.rank, self.sync_interval = world_size, rank, sync_interval
        self.which = tuple(which)
        self.device = device
        self._comm = None
        if world_size > 1:
            self._comm = self._connect(bcast_bytes)

    def _connect(self, bcast_bytes):
        L = _lib.lib()
        uid = (C.c_uint8 * _lib.BDR_UNIQUE_ID_BYTES)()
        if self.rank == 0:
            _lib.check(L.bdr_comm_get_unique_id(uid))
        raw = bcast_bytes(bytes(uid) if self.rank == 0 else None)
        uid = (C.c_uint8 * _lib.BDR_UNIQUE_ID_BYTES)(*raw)
        h = C.c_void_p()
        _lib.check(L.bdr_comm_init_rank(uid, self.world_size, self.rank, self.device, C.byref(h)))
        return h

    @classmethod
    def rccl_or_raise(cls, world_size: int, rank: int, sync_interval: int, device: int, bcast_bytes, which=("qnet",)):
        """The library's RCCL communicator on EVERY rank, or a RuntimeError on every rank: the outcome is agreed on by all
        ranks (MIN-reduce of a success flag over the torch.distributed control-plane group), so no rank is left waiting in a
        collective and no run continues on a slower data plane without saying so.  (Round 1 demoted silently to torch's RCCL
        and then to host staging; a multi-GPU measurement must not do that.)"""
        import torch
        import torch.distributed as dist
        ex, err = None, None
        try:
            ex = cls(world_size, rank, sync_interval, device, bcast_bytes, which)
        except Exception as e:  # noqa: BLE001  (reported below, on every rank)
            err = e
        ok = torch.tensor([0 if ex is None else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            if ex is not None:
                ex.close()
            raise RuntimeError(f"rank {rank}: the RCCL communicator could not be initialised on every rank ({err!r})")
        return ex

    def close(self):
        if self._comm:
            _lib.lib().bdr_comm_destroy(self._comm)
            self._comm = None

    def after_opt(self, agent, opt_steps: int) -> bool:
        """Called after every opt step; averages every sync_interval-th step."""
        if self.world_size == 1 or self.sync_interval <= 0 or opt_steps % self.sync_interval != 0:
            return False
        self.average(agent)
        return True

    def average(self, agent) -> None:
        if self.world_size == 1:
            return
        for w in self.which:
            _lib.check(_lib.lib().bdr_agent_allreduce_params(agent.handle, self._comm, agent.WHICH[w]))

    def agree(self, local_ok: bool = True) -> bool:
        """MIN over ranks of local_ok: every rank calls it before a collective (bdr_learner_ops::agree); a rank that failed
        passes False once and every rank learns it instead of blocking in the next all-reduce."""
        if self.world_size == 1:
            return bool(local_ok)
        out = C.c_int32(0)
        _lib.check(_lib.lib().bdr_comm_agree(self._comm, 1 if local_ok else 0, C.byref(out)))
        return bool(out.value)

    def broadcast(self, agent, root: int = 0) -> None:
        """The faithful learner->actors sync (SyncModel::sync_model on every actor)."""
        if self.world_size == 1:
            return
        for w in self.which:
            _lib.check(_lib.lib().bdr_agent_broadcast_params(agent.handle, self._comm, agent.WHICH[w], root))


def shard_seed(base_seed: int, rank: int) -> int:
    """Replay shard g draws from its own StdRng stream: seed + rank."""
    return base_seed + rank
