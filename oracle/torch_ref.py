"""ATen (PyTorch CPU) restatement of border-tch-agent's DQN opt step, op for op.

TEST INFRASTRUCTURE ONLY (same rule as border_oracle.c): used to generate the golden fixtures
under tests/golden/ (tests/golden/make_golden.py), to cross-check the C oracle, and as
bench.py's `cpu_baseline` (tch 0.16 is a thin FFI over these very ATen kernels, so this is the
closest obtainable stand-in for "border-tch-agent's CPU path": BASELINE.md section 2).

Each function cites the reference lines it follows.  Nothing here is imported by border_amd/.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

CNN_NAMES = ["c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias",
             "l1.weight", "l1.bias", "l2.weight", "l2.bias"]


def cnn_shapes(out_dim: int, n_stack: int = 4):
    """AtariCnn variable shapes (border-tch-agent/src/cnn/base.rs:23-36)."""
    return [(32, n_stack, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,),
            (512, 3136), (512,), (out_dim, 512), (out_dim,)]


def mlp_shapes(in_dim: int, units, out_dim: int):
    """Mlp variable shapes (border-tch-agent/src/mlp/base.rs:13-41), order ln0.w ln0.b ..."""
    shapes, i = [], in_dim
    for u in list(units) + [out_dim]:
        shapes += [(u, i), (u,)]
        i = u
    return shapes


def init_params(shapes, seed: int) -> np.ndarray:
    """Deterministic uniform(+-1/sqrt(fan_in)) init -> flat f32 vector in reference order.
    (The init scheme is irrelevant to parity: fixtures carry the weights.)"""
    rng = np.random.default_rng(seed)
    out, fan_in = [], 1
    for s in shapes:
        if len(s) > 1:
            fan_in = int(np.prod(s[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        out.append(rng.uniform(-bound, bound, size=s).astype(np.float32).ravel())
    return np.concatenate(out)


def unflatten(flat: np.ndarray, shapes):
    ts, o = [], 0
    for s in shapes:
        n = int(np.prod(s))
        ts.append(torch.from_numpy(np.array(flat[o:o + n], copy=True)).reshape(s))
        o += n
    return ts


def flatten(ts) -> np.ndarray:
    return np.concatenate([t.detach().numpy().ravel() for t in ts]).astype(np.float32)


def cnn_forward(p, x_u8: torch.Tensor) -> torch.Tensor:
    """cnn/base.rs:23-36.  x_u8: [B,n_stack,1,84,84] holding integers 0..255."""
    x = x_u8.squeeze(2).to(torch.float32) / 255  # :26
    x = F.conv2d(x, p[0], p[1], stride=4).relu()  # :27-28
    x = F.conv2d(x, p[2], p[3], stride=2).relu()  # :29-30
    x = F.conv2d(x, p[4], p[5], stride=1).relu().flatten(1)  # :31-32
    x = F.linear(x, p[6], p[7]).relu()  # :33-34
    return F.linear(x, p[8], p[9])  # :35


def mlp_forward(p, x: torch.Tensor, activation_out=False) -> torch.Tensor:
    """mlp/base.rs:13-41."""
    n = len(p) // 2
    for i in range(n):
        x = F.linear(x, p[2 * i], p[2 * i + 1])
        if i < n - 1 or activation_out:
            x = x.relu()
    return x


def optimizer_step(params, m, v, vmax, step, lr, adamw=None):
    """One libtorch optimizer step over `params` (grads in .grad) at step number `step` (1-based), in place: Adam::step with tch's
    nn::Adam::default() (opt.rs:35) when adamw is None, AdamW::step (adamw.cpp: decoupled decay `param.mul_(1 - lr * wd)` first;
    amsgrad: denominator from max_exp_avg_sq) for OptimizerConfig::AdamW = dict(beta1, beta2, wd, eps[, amsgrad]) (opt.rs:38-55)."""
    b1, b2, eps, wd, amsgrad = 0.9, 0.999, 1e-8, 0.0, False
    if adamw is not None:
        b1, b2, eps, wd = adamw["beta1"], adamw["beta2"], adamw["eps"], adamw["wd"]
        amsgrad = bool(adamw.get("amsgrad", False))
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    with torch.no_grad():
        for p, m_, v_, x_ in zip(params, m, v, vmax):
            g = p.grad
            if adamw is not None:
                p.mul_(1 - lr * wd)
            m_.mul_(b1).add_(g, alpha=1 - b1)
            v_.mul_(b2).addcmul_(g, g, value=1 - b2)
            if amsgrad:
                torch.maximum(x_, v_, out=x_)
                denom = (x_.sqrt() / math.sqrt(bc2)).add_(eps)
            else:
                denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m_, denom, value=-(lr / bc1))


class TorchDqn:
    """Dqn (dqn/base.rs) with DqnModel (dqn/model/base.rs) and tch's Adam (opt.rs:35)."""

    def __init__(self, kind, shapes, params: np.ndarray, *, lr, discount_factor=0.99, double_dqn=False,
                 critic_loss="Mse", clip_td_err=None, tau=0.005, soft_update_interval=1, adamw=None):
        self.kind, self.shapes = kind, shapes
        self.q = [t.requires_grad_(True) for t in unflatten(params, shapes)]
        self.q_tgt = unflatten(params, shapes)  # DqnModel::clone, dqn/model/base.rs:94-115
        self.m = [torch.zeros_like(t) for t in self.q]
        self.v = [torch.zeros_like(t) for t in self.q]
        self.lr, self.step = lr, 0
        self.gamma, self.double_dqn, self.critic_loss = discount_factor, double_dqn, critic_loss
        self.clip_td_err = clip_td_err
        self.adamw = adamw   # None (opt.rs:35 Adam) or dict(beta1, beta2, wd, eps[, amsgrad]) (opt.rs:38-55 AdamW)
        self.vmax = [torch.zeros_like(t) for t in self.q]   # max_exp_avg_sq (AdamW{amsgrad: true})
        self.tau, self.soft_update_interval, self.soft_update_counter = tau, soft_update_interval, 0

    def fwd(self, p, x):
        return cnn_forward(p, x) if self.kind == "cnn" else mlp_forward(p, x)

    def _adam(self):
        """libtorch Adam::step / AdamW::step (optimizer_step above)."""
        self.step += 1
        optimizer_step(self.q, self.m, self.v, self.vmax, self.step, self.lr, self.adamw)

    def update(self, obs, act, next_obs, reward, term, weight=None):
        """dqn/base.rs:60-160 followed by :190-196 (soft update)."""
        obs = torch.from_numpy(np.ascontiguousarray(obs))
        next_obs = torch.from_numpy(np.ascontiguousarray(next_obs))
        act = torch.from_numpy(np.ascontiguousarray(act, dtype=np.int64)).reshape(-1, 1)
        reward = torch.from_numpy(np.ascontiguousarray(reward, dtype=np.float32))
        is_terminated = torch.from_numpy(np.ascontiguousarray(term, dtype=np.int8))

        q_all = self.fwd(self.q, obs)
        pred = q_all.gather(-1, act).squeeze()  # :71-74
        with torch.no_grad():  # :91-105
            if self.double_dqn:
                x = self.fwd(self.q, next_obs)
                y = x.argmax(-1).unsqueeze(-1)
                qn_all = self.fwd(self.q_tgt, next_obs)
                q = qn_all.gather(-1, y).squeeze()
            else:
                qn_all = self.fwd(self.q_tgt, next_obs)
                y = qn_all.argmax(-1).unsqueeze(-1)
                q = qn_all.gather(-1, y).squeeze()
            tgt = reward + (1 - is_terminated) * self.gamma * q
        td_abs = None
        if weight is not None:  # :123-145
            n = len(weight)
            td = (pred - tgt).abs()
            if self.clip_td_err is not None:
                td = td.clip(*self.clip_td_err)
            td_abs = td.detach().numpy().copy()
            l = torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float32)) * td
            z = torch.zeros(n)
            loss = F.smooth_l1_loss(l, z, reduction="mean", beta=1.0) if self.critic_loss == "SmoothL1" \
                else F.mse_loss(l, z, reduction="mean")
        else:  # :146-152
            loss = F.smooth_l1_loss(pred, tgt, reduction="mean", beta=1.0) if self.critic_loss == "SmoothL1" \
                else F.mse_loss(pred, tgt, reduction="mean")
        # opt.rs:74-83 backward_step = zero_grad + backward + step
        for p in self.q:
            p.grad = None
        loss.backward()
        grads = flatten([p.grad for p in self.q])
        self._adam()
        # dqn/base.rs:190-196 + util.rs:31-45
        self.soft_update_counter += 1
        if self.soft_update_counter == self.soft_update_interval:
            self.soft_update_counter = 0
            with torch.no_grad():
                for d, s in zip(self.q_tgt, self.q):
                    d.copy_(self.tau * s + (1.0 - self.tau) * d)
        return dict(loss=float(loss.detach()), q_pred_all=q_all.detach().numpy().copy(),
                    q_next_all=qn_all.numpy().copy(), pred=pred.detach().numpy().copy(),
                    tgt=tgt.numpy().copy(), grads=grads, td_abs=td_abs)

    def params(self):
        return flatten(self.q)

    def tgt_params(self):
        return flatten(self.q_tgt)


def synthetic_atari_batch(B: int, A: int, seed: int):
    """SURVEY.md section 8(d) synthetic inputs for a fixed minibatch."""
    rng = np.random.default_rng(seed)
    obs = rng.integers(0, 256, size=(B, 4, 1, 84, 84), dtype=np.uint8)
    next_obs = rng.integers(0, 256, size=(B, 4, 1, 84, 84), dtype=np.uint8)
    act = rng.integers(0, A, size=(B,), dtype=np.int64)
    reward = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=B, p=[0.05, 0.9, 0.05]).astype(np.float32)
    term = (rng.random(B) < 0.005).astype(np.int8)
    return obs, act, next_obs, reward, term


def _timed(one, steps, warmup, seconds=None):
    """steps/s over `steps` steps - or, with `seconds`, over as many steps (>= 1, <= steps) as fit that much time."""
    import time
    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    n = 0
    while n < steps:
        one()
        n += 1
        if seconds is not None and time.perf_counter() - t0 >= seconds:
            break
    return n / (time.perf_counter() - t0), torch.get_num_threads()


_RINGS = {}


def _f32_ring(capacity, n_actions):
    """The reference's f32 observation ring (border-atari-env/src/obs.rs:45-52 stores frames as f32; tensor_batch.rs:95-120):
    [capacity,4,1,84,84] x 2.  Contents are a 1024-row random block tiled over the ring (what is timed is the gather's memory
    behaviour and the update, not the values); built once per process."""
    key = (capacity, n_actions)
    if key not in _RINGS:
        g = torch.Generator().manual_seed(0)
        blk = min(1024, capacity)
        block = torch.randint(0, 256, (blk, 4, 1, 84, 84), generator=g, dtype=torch.uint8).to(torch.float32)
        rings = []
        for shift in (0, 1):
            r = torch.empty((capacity, 4, 1, 84, 84), dtype=torch.float32)
            src = block.roll(shift, 0)
            for o in range(0, capacity, blk):
                n = min(blk, capacity - o)
                r[o:o + n].copy_(src[:n])
            rings.append(r)
        _RINGS[key] = (rings[0], rings[1], torch.randint(0, n_actions, (capacity, 1), generator=g), torch.zeros(capacity),
                       torch.zeros(capacity, dtype=torch.int8))
    return _RINGS[key]


def time_dqn_atari(batch_size=256, n_actions=6, steps=5, warmup=1, threads=None, capacity=65536,
                   critic_loss="SmoothL1", seconds=None):
    """CPU baseline: sample (index draw + 3x index_select on an f32 ring of 65 536 transitions, as the
    reference stores it) + update, timed like Trainer::train_step (border-core/src/trainer.rs:213-225).  Returns
    (opt-steps/sec, threads used)."""
    if threads:
        torch.set_num_threads(threads)
    shapes = cnn_shapes(n_actions)
    agent = TorchDqn("cnn", shapes, init_params(shapes, 0), lr=1e-4, critic_loss=critic_loss, tau=1.0,
                     soft_update_interval=10000)
    ring_obs, ring_next, ring_act, ring_rew, ring_term = _f32_ring(capacity, n_actions)
    rng = np.random.default_rng(42)

    def one():
        ixs = torch.from_numpy(rng.integers(0, capacity, batch_size))
        obs = ring_obs.index_select(0, ixs).numpy()
        nobs = ring_next.index_select(0, ixs).numpy()
        act = ring_act.index_select(0, ixs).numpy()
        agent.update(obs, act, nobs, ring_rew[ixs].numpy(), ring_term[ixs].numpy())

    return _timed(one, steps, warmup, seconds)


def time_dqn_cartpole(batch_size=32, steps=50, warmup=5, threads=None, capacity=10000, seconds=None):
    """BASELINE config 1 on the CPU path: Mlp[64,64], obs 4 f32, 2 actions, MSE, Adam 1e-3, tau 0.01 every opt."""
    if threads:
        torch.set_num_threads(threads)
    shapes = mlp_shapes(4, [64, 64], 2)
    agent = TorchDqn("mlp", shapes, init_params(shapes, 0), lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1)
    g = torch.Generator().manual_seed(0)
    obs, nobs = torch.randn(capacity, 4, generator=g), torch.randn(capacity, 4, generator=g)
    act, rew, term = torch.randint(0, 2, (capacity, 1), generator=g), torch.randn(capacity, generator=g), torch.zeros(capacity, dtype=torch.int8)
    rng = np.random.default_rng(42)

    def one():
        ixs = torch.from_numpy(rng.integers(0, capacity, batch_size))
        agent.update(obs.index_select(0, ixs).numpy(), act.index_select(0, ixs).numpy(), nobs.index_select(0, ixs).numpy(),
                     rew[ixs].numpy(), term[ixs].numpy())

    return _timed(one, steps, warmup, seconds)


def time_iqn_atari(batch_size=512, steps=2, warmup=1, threads=None, n_actions=6, n_quantiles=64, seconds=None):
    """BASELINE config 4 on the CPU path (fixed minibatch: the gather is negligible next to 0.5 TFLOP of update)."""
    if threads:
        torch.set_num_threads(threads)
    sh = iqn_shapes("cnn", 3136, 64, [512], n_actions)
    agent = TorchIqn("cnn", sh, init_params(sh[0] + sh[1] + sh[2], 0), lr=1e-4, feature_dim=3136, embed_dim=64, tau=1.0, soft_update_interval=10000)
    batch = iqn_batch(batch_size, "cnn", n_actions, n_quantiles, n_quantiles, 1)
    return _timed(lambda: agent.update(*batch), steps, warmup, seconds)


def time_sac(batch_size=1024, steps=20, warmup=2, threads=None, obs_dim=17, act_dim=6, seconds=None):
    """BASELINE config 5 on the CPU path: twin-Q [256,256], actor [256,256], Auto entropy coefficient."""
    if threads:
        torch.set_num_threads(threads)
    pu, qu = [256, 256], [256, 256]
    pi0 = init_params(sac_pi_shapes(obs_dim, pu, act_dim), 1) * np.float32(0.5)
    q0 = [init_params(sac_q_shapes(obs_dim, act_dim, qu), 2 + i) for i in range(2)]
    agent = TorchSac(obs_dim, act_dim, pu, qu, pi0, q0, lr_actor=3e-4, lr_critic=3e-4, ent_coef=("Auto", -6.0, 3e-4))
    batch = sac_batch(batch_size, obs_dim, act_dim, 3)
    return _timed(lambda: agent.update(*batch), steps, warmup, seconds)


def time_c_oracle_dqn(batch_size=256, n_actions=6, critic_loss="SmoothL1"):
    """The scalar C restatement (oracle/border_oracle.c, OpenMP over its outer loops; double accumulators) on the same step,
    for context: all threads on the full batch, one thread on 1/16 of the batch (scaled)."""
    import os
    import time
    from oracle import oracle as O
    shapes = cnn_shapes(n_actions)
    p0 = init_params(shapes, 0)

    def run(B, reps):
        ref = O.DqnOracle(O.cnn_cfg(n_actions), p0, lr=1e-4, critic_loss=critic_loss, tau=1.0, soft_update_interval=10000)
        b = synthetic_atari_batch(B, n_actions, 5)
        ref.update(*b)
        t0 = time.perf_counter()
        for _ in range(reps):
            ref.update(*b)
        return reps / (time.perf_counter() - t0)

    nthreads = int(O.lib().orc_num_threads())
    all_t = run(batch_size, 2)
    old = os.environ.get("OMP_NUM_THREADS")
    O.lib().orc_set_num_threads(1)
    small = max(1, batch_size // 16)
    one_t = run(small, 1) * small / batch_size
    O.lib().orc_set_num_threads(nthreads)
    if old is not None:
        os.environ["OMP_NUM_THREADS"] = old
    return {"opt_steps_per_s_all_threads": round(all_t, 4), "threads": nthreads,
            "opt_steps_per_s_1_thread": round(one_t, 5), "one_thread_sample": f"batch {small}, scaled to {batch_size}"}


# ================================================================================================
# SAC  (border-tch-agent/src/sac/base.rs:73-198, mlp/mlp2.rs:23-50, sac/ent_coef.rs:27-75)
# ================================================================================================
def sac_pi_shapes(obs_dim, units, act_dim):
    """Actor (Mlp2) variables: mlp.al{i}.weight/bias, ml.weight/bias, sl.weight/bias."""
    shapes, i = [], obs_dim
    for u in units:
        shapes += [(u, i), (u,)]
        i = u
    return shapes + [(act_dim, i), (act_dim,), (act_dim, i), (act_dim,)]


def sac_q_shapes(obs_dim, act_dim, units):
    """Critic (Mlp as SubModel2, mlp/base.rs:83-107): cat(obs, act) -> units -> 1."""
    return mlp_shapes(obs_dim + act_dim, units, 1)


class TorchSac:
    """Sac::opt_ with injected N(0,1) noise (the reference draws it from torch's global CPU
    generator: sac/base.rs:76) so a fixed (minibatch, z_actor, z_next) is reproducible."""

    def __init__(self, obs_dim, act_dim, pi_units, q_units, pi_params, q_params_list, *, lr_actor, lr_critic, gamma=0.99,
                 tau=0.005, ent_coef=("Fix", 1.0), epsilon=1e-4, min_lstd=-20.0, max_lstd=2.0, reward_scale=1.0,
                 critic_loss="Mse", adamw_actor=None, adamw_critic=None):
        self.pi_shapes = sac_pi_shapes(obs_dim, pi_units, act_dim)
        self.q_shapes = sac_q_shapes(obs_dim, act_dim, q_units)
        self.n_trunk = len(pi_units)
        self.pi = [t.requires_grad_(True) for t in unflatten(pi_params, self.pi_shapes)]
        self.qs = [[t.requires_grad_(True) for t in unflatten(p, self.q_shapes)] for p in q_params_list]
        self.qs_tgt = [unflatten(p, self.q_shapes) for p in q_params_list]   # Critic::clone
        self.gamma, self.tau, self.eps = gamma, tau, epsilon
        self.min_lstd, self.max_lstd, self.reward_scale, self.critic_loss = min_lstd, max_lstd, reward_scale, critic_loss
        if ent_coef[0] == "Fix":   # ent_coef.rs:33-38
            self.log_alpha = torch.tensor([math.log(ent_coef[1])], dtype=torch.float32)
            self.target_entropy, self.lr_alpha = None, None
        else:                      # Auto(target_entropy, lr): ent_coef.rs:39-46
            self.log_alpha = torch.zeros(1, requires_grad=True)
            self.target_entropy, self.lr_alpha = ent_coef[1], ent_coef[2]
        # ActorConfig / CriticConfig.opt_config (Adam, or AdamW = dict(beta1, beta2, wd, eps[, amsgrad])); EntCoef: nn::Adam::default() (ent_coef.rs:41)
        self.opt = {"pi": self._state(self.pi, lr_actor, adamw_actor), "alpha": self._state([self.log_alpha], self.lr_alpha or 0.0)}
        for i, q in enumerate(self.qs):
            self.opt[f"q{i}"] = self._state(q, lr_critic, adamw_critic)

    @staticmethod
    def _state(params, lr, adamw=None):
        return dict(m=[torch.zeros_like(p) for p in params], v=[torch.zeros_like(p) for p in params],
                    vmax=[torch.zeros_like(p) for p in params], step=0, lr=lr, adamw=adamw)

    @staticmethod
    def _adam(params, st):
        st["step"] += 1
        optimizer_step(params, st["m"], st["v"], st["vmax"], st["step"], st["lr"], st["adamw"])

    def pi_forward(self, o):
        """Mlp2::forward (mlp2.rs:23-28): returns (mean, exp(head2))."""
        x = o
        for i in range(self.n_trunk):
            x = F.linear(x, self.pi[2 * i], self.pi[2 * i + 1]).relu()
        k = 2 * self.n_trunk
        return F.linear(x, self.pi[k], self.pi[k + 1]), F.linear(x, self.pi[k + 2], self.pi[k + 3]).exp()

    def action_logp(self, o, z):
        """sac/base.rs:73-87 (note: `lstd` is already exp(head2) and is exponentiated again)."""
        mean, lstd = self.pi_forward(o)
        std = lstd.clip(self.min_lstd, self.max_lstd).exp()
        a = (std * z + mean).tanh()
        normal_logp = (torch.tensor(-0.5 * math.log(2.0 * math.pi), dtype=torch.float32) - 0.5 * z.pow(2)).sum(-1)
        log_p = normal_logp - (torch.tensor(1.0) - a.pow(2.0) + torch.tensor(self.eps)).log().sum(-1)
        return a, log_p

    def alpha(self):
        return self.log_alpha.detach().exp()

    @staticmethod
    def q_forward(q, o, a):
        return mlp_forward(q, torch.cat([o, a], -1)).squeeze()

    def update(self, obs, act, next_obs, reward, term, z_actor, z_next):
        o = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32))
        act = torch.from_numpy(np.ascontiguousarray(act, dtype=np.float32))
        no = torch.from_numpy(np.ascontiguousarray(next_obs, dtype=np.float32))
        reward = torch.from_numpy(np.ascontiguousarray(reward, dtype=np.float32))
        is_terminated = torch.from_numpy(np.ascontiguousarray(term, dtype=np.int8))
        z_actor = torch.from_numpy(np.ascontiguousarray(z_actor, dtype=np.float32))
        z_next = torch.from_numpy(np.ascontiguousarray(z_next, dtype=np.float32))
        out = {}
        # ---- update_actor (sac/base.rs:151-167), first (:181)
        a, log_p = self.action_logp(o, z_actor)
        if self.target_entropy is not None:   # ent_coef.rs:69-75
            loss_a = -(self.log_alpha * (log_p.detach() + torch.tensor(self.target_entropy))).mean()
            self.log_alpha.grad = None
            loss_a.backward()
            self._adam([self.log_alpha], self.opt["alpha"])
        qmin = torch.vstack([self.q_forward(q, o, a) for q in self.qs]).min(0)[0]
        loss_actor = (self.alpha() * log_p - qmin).mean()
        for p in self.pi:
            p.grad = None
        loss_actor.backward()
        out["pi_grads"] = flatten([p.grad for p in self.pi])
        out["a"], out["log_p"] = a.detach().numpy().copy(), log_p.detach().numpy().copy()
        self._adam(self.pi, self.opt["pi"])
        # ---- update_critic (:107-149)
        preds = [self.q_forward(q, o, act) for q in self.qs]
        with torch.no_grad():
            next_a, next_log_p = self.action_logp(no, z_next)
            next_q = torch.vstack([self.q_forward(q, no, next_a) for q in self.qs_tgt]).min(0)[0]
            next_q = next_q - self.alpha() * next_log_p
            tgt = self.reward_scale * reward + (1.0 - is_terminated) * torch.tensor(self.gamma) * next_q
        losses = [F.mse_loss(p, tgt) if self.critic_loss == "Mse" else F.smooth_l1_loss(p, tgt, beta=1.0) for p in preds]
        out["q_grads"] = []
        for i, (q, l) in enumerate(zip(self.qs, losses)):
            for p in q:
                p.grad = None
            l.backward()
            out["q_grads"].append(flatten([p.grad for p in q]))
            self._adam(q, self.opt[f"q{i}"])
        # ---- soft_update (:169-173), every update
        with torch.no_grad():
            for qt, q in zip(self.qs_tgt, self.qs):
                for d, s in zip(qt, q):
                    d.copy_(self.tau * s + (1.0 - self.tau) * d)
        out.update(loss_critic=float(sum(float(l.detach()) for l in losses) / len(losses)), loss_actor=float(loss_actor.detach()),
                   ent_coef=float(self.alpha()[0]), tgt=tgt.numpy().copy(), preds=[p.detach().numpy().copy() for p in preds],
                   pi_params=flatten(self.pi), q_params=[flatten(q) for q in self.qs],
                   q_tgt_params=[flatten(q) for q in self.qs_tgt], log_alpha=float(self.log_alpha.detach()[0]))
        return out


def sac_batch(B, obs_dim, act_dim, seed):
    rng = np.random.default_rng(seed)
    obs = rng.standard_normal((B, obs_dim)).astype(np.float32)
    nobs = rng.standard_normal((B, obs_dim)).astype(np.float32)
    act = rng.uniform(-1, 1, (B, act_dim)).astype(np.float32)
    rew = rng.standard_normal(B).astype(np.float32)
    term = (rng.random(B) < 0.05).astype(np.int8)
    z1 = rng.standard_normal((B, act_dim)).astype(np.float32)
    z2 = rng.standard_normal((B, act_dim)).astype(np.float32)
    return obs, act, nobs, rew, term, z1, z2


# ================================================================================================
# IQN  (border-tch-agent/src/iqn/base.rs:63-170, iqn/model/base.rs:162-234, util/quantile_loss.rs:7-13)
# ================================================================================================
def iqn_shapes(psi_kind, feature_dim, embed_dim, f_units, n_actions, psi_in=None, psi_units=(), n_stack=4):
    """IqnModel variables: psi (AtariCnn{skip_linear:true}: c1..c3 | Mlp in->units->feature_dim),
    iqn_cos_to_feature.weight/bias, f = Mlp(feature_dim -> f_units -> n_actions)."""
    if psi_kind == "cnn":
        psi = [(32, n_stack, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,)]
        assert feature_dim == 3136
    else:
        psi = mlp_shapes(psi_in, psi_units, feature_dim)
    return psi, [(feature_dim, embed_dim), (feature_dim,)], mlp_shapes(feature_dim, f_units, n_actions)


def quantile_huber_loss(x, tau):
    """util/quantile_loss.rs:7-13."""
    lt_0 = x.lt(0.0).detach()
    loss = F.smooth_l1_loss(x, torch.zeros_like(x), reduction="none", beta=1.0)
    return (tau - torch.where(lt_0, 1.0, 0.0)).abs() * loss


class TorchIqn:
    """Iqn::update_critic / opt_ with injected percent points (the reference draws them with
    Tensor::rand on the CPU generator: iqn/model/base.rs:365-368)."""

    def __init__(self, psi_kind, shapes3, params, *, lr, feature_dim, embed_dim, discount_factor=0.99, tau=0.005,
                 soft_update_interval=1, psi_activation_out=True, adamw=None):
        self.psi_kind = psi_kind
        self.adamw = adamw   # IqnModelConfig.opt_config: None = Adam{lr}, or AdamW = dict(beta1, beta2, wd, eps[, amsgrad])
        self.shapes = shapes3[0] + shapes3[1] + shapes3[2]
        self.n_psi, self.n_f = len(shapes3[0]), len(shapes3[2])
        self.p = [t.requires_grad_(True) for t in unflatten(params, self.shapes)]
        self.p_tgt = unflatten(params, self.shapes)   # IqnModel::clone
        self.m = [torch.zeros_like(t) for t in self.p]
        self.v = [torch.zeros_like(t) for t in self.p]
        self.vmax = [torch.zeros_like(t) for t in self.p]
        self.lr, self.step, self.gamma, self.tau = lr, 0, discount_factor, tau
        self.F, self.E = feature_dim, embed_dim
        self.soft_update_interval, self.soft_update_counter = soft_update_interval, 0
        self.psi_activation_out = psi_activation_out

    def forward(self, p, x, tau):
        """IqnModel::forward (iqn/model/base.rs:198-234)."""
        psi_p, cos_p, f_p = p[:self.n_psi], p[self.n_psi:self.n_psi + 2], p[self.n_psi + 2:]
        if self.psi_kind == "cnn":   # AtariCnn::create_net_wo_linear (cnn/base.rs:38-47)
            h = x.squeeze(2).to(torch.float32) / 255
            h = F.conv2d(h, psi_p[0], psi_p[1], stride=4).relu()
            h = F.conv2d(h, psi_p[2], psi_p[3], stride=2).relu()
            psi = F.conv2d(h, psi_p[4], psi_p[5], stride=1).relu().flatten(1)
        else:
            psi = mlp_forward(psi_p, x, activation_out=self.psi_activation_out)
        B, N = tau.shape
        i = torch.arange(1, self.E + 1, dtype=torch.float32).reshape(1, 1, -1)        # Tensor::range(1, embed_dim) inclusive
        cos = torch.cos(tau.unsqueeze(-1) * (math.pi * i)).reshape(-1, self.E)         # :162-182
        phi = F.linear(cos, cos_p[0], cos_p[1]).relu().reshape(B, N, self.F)            # iqn_cos_to_feature + relu
        m = psi.unsqueeze(1) * phi
        return mlp_forward(f_p, m)                                                     # [B, N, A]

    def update(self, obs, act, next_obs, reward, term, tau_pred, tau_tgt):
        """iqn/base.rs:63-170 + opt_ :172-191."""
        obs = torch.from_numpy(np.ascontiguousarray(obs))
        next_obs = torch.from_numpy(np.ascontiguousarray(next_obs))
        act = torch.from_numpy(np.ascontiguousarray(act, dtype=np.int64)).reshape(-1, 1)
        reward = torch.from_numpy(np.ascontiguousarray(reward, dtype=np.float32)).unsqueeze(-1)
        is_terminated = torch.from_numpy(np.ascontiguousarray(term, dtype=np.int8)).unsqueeze(-1)
        tau_p = torch.from_numpy(np.ascontiguousarray(tau_pred, dtype=np.float32))
        tau_t = torch.from_numpy(np.ascontiguousarray(tau_tgt, dtype=np.float32))
        n_p, n_t = tau_p.shape[1], tau_t.shape[1]
        z = self.forward(self.p, obs, tau_p)
        a = act.unsqueeze(1).repeat(1, n_p, 1)
        pred = z.gather(-1, a).squeeze(-1).unsqueeze(1)          # [B,1,Np]
        with torch.no_grad():
            zt = self.forward(self.p_tgt, next_obs, tau_t)
            y = zt.clone().mean(1)
            a2 = y.argmax(-1).unsqueeze(-1).unsqueeze(-1).repeat(1, n_t, 1)
            zsel = zt.gather(2, a2).squeeze(-1)
            tgt = (reward + (1 - is_terminated) * self.gamma * zsel).unsqueeze(-1)   # [B,Nt,1]
        diff = tgt - pred
        tau_rep = tau_p.unsqueeze(1).repeat(1, n_t, 1)
        loss = quantile_huber_loss(diff, tau_rep).mean()
        for p in self.p:
            p.grad = None
        loss.backward()
        grads = flatten([p.grad for p in self.p])
        self.step += 1
        optimizer_step(self.p, self.m, self.v, self.vmax, self.step, self.lr, self.adamw)
        self.soft_update_counter += 1
        if self.soft_update_counter == self.soft_update_interval:
            self.soft_update_counter = 0
            with torch.no_grad():
                for d, s in zip(self.p_tgt, self.p):
                    d.copy_(self.tau * s + (1.0 - self.tau) * d)
        return dict(loss=float(loss.detach()), z_pred=z.detach().numpy().copy(), z_tgt=zt.numpy().copy(),
                    tgt=tgt.squeeze(-1).numpy().copy(), grads=grads, params=flatten(self.p), tgt_params=flatten(self.p_tgt))


def iqn_batch(B, psi_kind, n_actions, n_pred, n_tgt, seed, in_dim=None, n_stack=4):
    rng = np.random.default_rng(seed)
    if psi_kind == "cnn":
        obs = rng.integers(0, 256, size=(B, n_stack, 1, 84, 84), dtype=np.uint8)
        nobs = rng.integers(0, 256, size=(B, n_stack, 1, 84, 84), dtype=np.uint8)
    else:
        obs = rng.standard_normal((B, in_dim)).astype(np.float32)
        nobs = rng.standard_normal((B, in_dim)).astype(np.float32)
    act = rng.integers(0, n_actions, size=B)
    rew = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=B).astype(np.float32)
    term = (rng.random(B) < 0.1).astype(np.int8)
    return obs, act, nobs, rew, term, rng.random((B, n_pred), dtype=np.float32), rng.random((B, n_tgt), dtype=np.float32)
