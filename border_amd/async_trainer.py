"""Host-side mirror of border-async-trainer over the C ABI (csrc/async_trainer.hip).

  AsyncTrainerConfig   border-async-trainer/src/async_trainer/config.rs:13-36 (defaults :101-112)
  ActorManagerConfig   actor_manager/config.rs:5-16 (n_buffer = 100)
  AsyncTrainer.train   train_async (util.rs:31-92) = ActorManager::run + AsyncTrainer::train + stop_and_join:
                       the loops themselves run in compiled code (bdr_async_train: one learner on the calling thread, one
                       std::thread per actor); Python supplies the environments (callbacks) and, optionally, an observer.
  ModelMailbox         the device-resident SyncModel::ModelInfo slot that replaces the learner -> actors channel.
  AsyncTrainStat / ActorStat   async_trainer/stat.rs, actor/stat.rs

Multi-GPU (north_star): every rank runs AsyncTrainer.train with its own learner, actors and replay shard; `exchange`
(a ParamExchange) averages the learners' parameters over RCCL at every sync point before the local publish.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class AsyncTrainerConfig:
    max_opts: int = 10
    eval_interval: int = 5000               # (evaluator: out of scope)
    flush_record_interval: int = 5000       # (recorder: out of scope)
    record_compute_cost_interval: int = 5000
    record_agent_info_interval: int = 5000
    save_interval: int = 50000              # (recorder: out of scope)
    sync_interval: int = 100
    warmup_period: int = 10000
    warmup_sleep_ms: int = 100              # async_trainer/base.rs:331


@dataclass
class ActorManagerConfig:
    n_buffer: int = 100
    channel_capacity: int = 1000            # bounded(1000), actor_manager/base.rs:140


@dataclass
class AsyncTrainStat:
    samples_per_sec: float
    duration: float
    opt_per_sec: float
    samples_total: int = 0
    opt_steps: int = 0
    n_syncs: int = 0
    n_messages: int = 0
    n_records: int = 0

    def fmt(self) -> str:                   # async_trainer/stat.rs:15-26
        return f"samples/sec, opt_steps/sec, duration\n{self.samples_per_sec}, {self.opt_per_sec}, {self.duration}\n"


@dataclass
class ActorStat:
    env_steps: int
    duration: float
    n_syncs: int = 0


def actor_stats_fmt(stats) -> str:          # actor/stat.rs:14-23
    s = "actor_id, samples, samples/sec, duration\n"
    for i, st in enumerate(stats):
        s += f"{i}, {st.env_steps}, {st.env_steps / max(st.duration, 1e-12)}, {st.duration}\n"
    return s


class AsyncTrainerConfigC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("max_opts", "warmup_period", "sync_interval", "record_agent_info_interval",
                                          "record_compute_cost_interval", "n_buffer", "channel_capacity", "warmup_sleep_ms",
                                          "obs_row_bytes", "act_row_bytes")]


LEN_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64))
PUBLISH_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64)
AGREE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32))
SYNC_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_int32))
ASYNC_OBSERVER_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_float), C.c_int32)


class LearnerOps(C.Structure):
    _fields_ = [("t", _lib.TrainerOps), ("buffer_len", LEN_FN), ("publish_model", PUBLISH_FN), ("mailbox", C.c_void_p),
                ("exchange", EXCHANGE_FN), ("exchange_ctx", C.c_void_p), ("agree", AGREE_FN)]


class ActorOps(C.Structure):
    _fields_ = [("agent", C.c_void_p), ("mailbox", C.c_void_p), ("agent_set_train", _lib.SET_TRAIN_FN), ("agent_sample", _lib.SAMPLE_FN),
                ("sync_model", SYNC_FN), ("agent_sample_device", _lib.SAMPLE_DEV_FN), ("env", _lib.EnvVtable)]


class AsyncStatsC(C.Structure):
    _fields_ = [("samples_total", C.c_uint64), ("opt_steps", C.c_uint64), ("n_records", C.c_uint64), ("n_syncs", C.c_uint64),
                ("n_messages", C.c_uint64), ("duration_s", C.c_double), ("samples_per_sec", C.c_float), ("opt_per_sec", C.c_float)]


class ActorStatC(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("n_syncs", C.c_uint64), ("duration_s", C.c_double)]


EVENTS = {0: "skip", 1: "opt", 2: "opt_record", 3: "cost", 4: "sync", 5: "push", 6: "actor_sync"}
LEARNER = 0xFFFFFFFF


def _bind():
    L = _lib.lib()
    L.bdr_model_mailbox_create.argtypes = [C.c_int32, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]
    L.bdr_model_mailbox_destroy.argtypes = [C.c_void_p]
    L.bdr_agent_publish_model.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]
    L.bdr_agent_sync_model_from.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    L.bdr_async_trainer_config_default.argtypes = [C.POINTER(AsyncTrainerConfigC)]
    L.bdr_async_trainer_config_default.restype = None
    L.bdr_learner_ops_default.argtypes = [C.POINTER(LearnerOps), C.c_void_p, C.c_void_p, C.c_void_p]
    L.bdr_learner_ops_default.restype = None
    L.bdr_actor_ops_default.argtypes = [C.POINTER(ActorOps), C.c_void_p, C.c_void_p, C.POINTER(_lib.EnvVtable)]
    L.bdr_actor_ops_default.restype = None
    L.bdr_async_train.argtypes = [C.POINTER(AsyncTrainerConfigC), C.POINTER(LearnerOps), C.POINTER(ActorOps), C.c_uint32, ASYNC_OBSERVER_FN,
                                  C.c_void_p, C.POINTER(AsyncStatsC), C.POINTER(ActorStatC)]
    return L


class ModelMailbox:
    """Device-resident (n_opts, parameters) slot between one learner and its actors on the same GPU."""

    def __init__(self, agent, n_readers: int, which: str = None, device: int = 0):
        which = which or next(iter(agent.WHICH))
        _, n = agent.arena_device_ptr(which)      # only the size is wanted: hand the arena straight back, or the learner would re-split
        _lib.check(_lib.lib().bdr_agent_arena_release(agent.handle, agent.WHICH[which]))   # its weight planes before every forward from here on
        h = C.c_void_p()
        _lib.check(_bind().bdr_model_mailbox_create(device, n, n_readers, C.byref(h)))
        self._h, self.which = h, agent.WHICH[which]

    @property
    def handle(self):
        return self._h

    def publish(self, agent, n_opts: int):
        _lib.check(_bind().bdr_agent_publish_model(agent.handle, self.which, self._h, n_opts))

    def sync(self, agent, reader: int, n_opts: int, first: bool = False):
        """-> (n_opts of the agent's model afterwards, updated?)"""
        v, up = C.c_uint64(n_opts), C.c_int32()
        _lib.check(_bind().bdr_agent_sync_model_from(agent.handle, self.which, self._h, reader, int(first), C.byref(v), C.byref(up)))
        return v.value, bool(up.value)

    def close(self):
        if getattr(self, "_h", None):
            _bind().bdr_model_mailbox_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def env_vtable(env, obs_shape, obs_dtype, act_row_bytes=8, act_dtype=np.int64, keep=None):
    """ctypes view of a Python environment (reset(None) -> obs[1, ...], step_with_reset(act) -> Step).  Exceptions inside a
    callback cannot cross the C frame: they are turned into a non-zero status (the loop stops and reports it)."""
    obs_dtype = np.dtype(obs_dtype)
    row = int(np.prod(obs_shape)) * obs_dtype.itemsize

    def write(ptr, arr):
        C.memmove(ptr, np.ascontiguousarray(arr, obs_dtype).ctypes.data, row)

    def reset(_ctx, obs_out):
        try:
            write(obs_out, env.reset(None))
            return 0
        except Exception:  # noqa: BLE001
            return 90

    def step(_ctx, act, obs_out, reward, term, trunc, init_out):
        try:
            a = np.frombuffer((C.c_char * act_row_bytes).from_address(act), act_dtype).copy()
            st = env.step_with_reset(a)
            write(obs_out, st.obs)
            reward[0], term[0], trunc[0] = float(st.reward[0]), int(st.is_terminated[0]), int(st.is_truncated[0])
            if st.is_done():
                write(init_out, st.init_obs)
            return 0
        except Exception:  # noqa: BLE001
            return 91

    if getattr(env, "device_obs", False):
        # device-resident observations (bdr_env_vtable::obs_on_device): obs_out / init_out are device buffers the environment
        # fills itself (AtariDeviceEnv: a copy of its frame stack inside HBM)
        def reset(_ctx, obs_out):   # noqa: F811
            try:
                env.reset_into(obs_out)
                return 0
            except Exception:  # noqa: BLE001
                return 90

        def step(_ctx, act, obs_out, reward, term, trunc, init_out):   # noqa: F811
            try:
                a = np.frombuffer((C.c_char * act_row_bytes).from_address(act), act_dtype).copy()
                reward[0], term[0], trunc[0] = env.step_into(a, obs_out, init_out)
                return 0
            except Exception:  # noqa: BLE001
                return 91

    fns = (_lib.ENV_RESET_FN(reset), _lib.ENV_STEP_FN(step))
    if keep is not None:
        keep.append(fns)
    vt = _lib.EnvVtable(None, *fns)
    if getattr(env, "device_obs", False):
        vt.obs_on_device, vt.device = 1, int(getattr(env, "device", 0))
    return vt


class AsyncTrainer:
    """train_async (util.rs:31-92) on one GPU: the learner `agent` over `buffer`, `actor_agents[i]` in `envs[i]`."""

    def __init__(self, config: AsyncTrainerConfig, actor_man_config: ActorManagerConfig = None):
        self.config, self.actor_man_config = config, actor_man_config or ActorManagerConfig()
        self.stat, self.actor_stats = None, None

    def _config(self, obs_row_bytes, act_row_bytes):
        c = AsyncTrainerConfigC()
        _bind().bdr_async_trainer_config_default(C.byref(c))
        k, m = self.config, self.actor_man_config
        c.max_opts, c.warmup_period, c.sync_interval = k.max_opts, k.warmup_period, k.sync_interval
        c.record_agent_info_interval, c.record_compute_cost_interval = k.record_agent_info_interval, k.record_compute_cost_interval
        c.n_buffer, c.channel_capacity, c.warmup_sleep_ms = m.n_buffer, m.channel_capacity, k.warmup_sleep_ms
        c.obs_row_bytes, c.act_row_bytes = obs_row_bytes, act_row_bytes
        return c

    def train(self, agent, buffer, actor_agents, envs, obs_shape, obs_dtype, act_row_bytes=8, on_event=None, exchange=None,
              learner_ops=None, actor_ops=None, mailbox=None, act_dtype=np.int64, agree=None):
        """Runs until the learner has done max_opts opt steps.  `exchange(opt_steps)`: optional cross-rank hook called at every
        sync point (e.g. lambda s: param_exchange.average(agent)).  `agree(local_ok) -> bool`: optional agreement across ranks
        before every exchange (ParamExchange.agree): a rank whose learner or actor fails says so once and every rank stops at
        the same sync point instead of blocking in a collective its peer never joins.  learner_ops / actor_ops: pre-built function tables (mock
        objects in the CPU tests); default = the library's handles with a device mailbox."""
        L = _bind()
        keep = []
        n_act = len(envs)
        row = int(np.prod(obs_shape)) * np.dtype(obs_dtype).itemsize
        own_mailbox = None
        if learner_ops is None:
            if mailbox is None:
                own_mailbox = mailbox = ModelMailbox(agent, n_act, device=getattr(buffer, "device", 0))
            learner_ops = LearnerOps()
            L.bdr_learner_ops_default(C.byref(learner_ops), agent.handle, buffer.handle, mailbox.handle)
        if exchange is not None:
            def exch(_ctx, _agent, opt_steps):
                try:
                    exchange(opt_steps)
                    return 0
                except Exception:  # noqa: BLE001
                    return 92
            fn = EXCHANGE_FN(exch)
            keep.append(fn)
            learner_ops.exchange = fn
        if agree is not None:
            def agr(_ctx, local_ok, all_ok):
                try:
                    all_ok[0] = 1 if agree(bool(local_ok)) else 0
                    return 0
                except Exception:  # noqa: BLE001
                    return 93
            fn = AGREE_FN(agr)
            keep.append(fn)
            learner_ops.agree = fn
        if actor_ops is None:
            if mailbox is None:
                raise ValueError("AsyncTrainer.train: the default actor function tables need the learner's ModelMailbox "
                                 "(pass mailbox=... together with learner_ops=...)")
            actor_ops = (ActorOps * n_act)()
            for i in range(n_act):
                vt = env_vtable(envs[i], obs_shape, obs_dtype, act_row_bytes, act_dtype, keep=keep)
                L.bdr_actor_ops_default(C.byref(actor_ops[i]), actor_agents[i].handle, mailbox.handle, C.byref(vt))
        else:
            arr = (ActorOps * n_act)()
            for i in range(n_act):
                arr[i] = actor_ops[i]
            actor_ops = arr

        # No observer, no callback: the compiled loop must not enter Python (and take the GIL, under its observer mutex, beside
        # the actors' environment callbacks) on every iteration, message and sync when nobody listens.
        cb = ASYNC_OBSERVER_FN()   # NULL function pointer
        if on_event is not None:
            def obs_cb(_ctx, actor, a, b, event, scalars, n):
                vals = [scalars[i] for i in range(n)] if event in (2, 3) else n
                on_event(None if actor == LEARNER else actor, a, b, EVENTS[event], vals)
            cb = ASYNC_OBSERVER_FN(obs_cb)
        c, st, ast = self._config(row, act_row_bytes), AsyncStatsC(), (ActorStatC * n_act)()
        rc = L.bdr_async_train(C.byref(c), C.byref(learner_ops), actor_ops, n_act, cb, None, C.byref(st), ast)
        # the loop's own status is reported FIRST (a device error found by the clean-up below must not mask it), and the
        # mailbox is closed whatever happens
        run_err = None
        try:
            _lib.check(rc)
        except Exception as e:  # noqa: BLE001
            run_err = e
        try:
            if own_mailbox is not None:
                agent.sync()
                for a in actor_agents:
                    a.sync()
        except Exception:  # noqa: BLE001
            if run_err is None:
                raise
        finally:
            if own_mailbox is not None:
                own_mailbox.close()
            if run_err is not None:
                raise run_err
        self.stat = AsyncTrainStat(st.samples_per_sec, st.duration_s, st.opt_per_sec, st.samples_total, st.opt_steps, st.n_syncs,
                                   st.n_messages, st.n_records)
        self.actor_stats = [ActorStat(ast[i].env_steps, ast[i].duration_s, ast[i].n_syncs) for i in range(n_act)]
        return self.stat
