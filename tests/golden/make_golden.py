"""Generates the golden fixtures in this directory.  Run once, in the build container:

    python tests/golden/make_golden.py

Sources of truth
  * rng_kat.json   -- rand 0.8.5's own known-answer vector (test_stdrng_construction) and the
                      ChaCha20 zero-key block, written out as literals; plus StdRng::seed_from_u64(42)
                      words / index batches produced by an independent pure-Python ChaCha12 below
                      (NOT by oracle/border_oracle.c, so the C oracle is checked against it).
  * dqn_*.npz      -- PyTorch CPU (oracle/torch_ref.py: the ATen ops tch 0.16 binds) run on seeded
                      inputs.  Large tensors (CNN weights, gradients, parameters) are stored as a
                      seed + strided samples + per-variable norms to keep the fixtures small.
The reference (Rust) cannot run here, and holds no golden vectors for this path.
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

M = 0xFFFFFFFF


def _rotl(x, n):
    return ((x << n) & M) | (x >> (32 - n))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & M; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & M; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & M; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & M; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key, counter, rounds):
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key) + [counter & M, (counter >> 32) & M, 0, 0]
    w = list(st)
    for _ in range(rounds // 2):
        _qr(w, 0, 4, 8, 12); _qr(w, 1, 5, 9, 13); _qr(w, 2, 6, 10, 14); _qr(w, 3, 7, 11, 15)
        _qr(w, 0, 5, 10, 15); _qr(w, 1, 6, 11, 12); _qr(w, 2, 7, 8, 13); _qr(w, 3, 4, 9, 14)
    return [(w[i] + st[i]) & M for i in range(16)]


class PyStdRng:
    def __init__(self, seed32):
        self.key = struct.unpack("<8I", bytes(seed32))
        self.pos = 0
        self._blk, self._blk_no = None, -1

    def u32(self):
        b = self.pos // 16
        if b != self._blk_no:
            self._blk, self._blk_no = chacha_block(self.key, b, 12), b
        v = self._blk[self.pos % 16]
        self.pos += 1
        return v


def seed_from_u64(state):
    out = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
        xs = (((state >> 18) ^ state) >> 27) & M
        rot = state >> 59
        x = ((xs >> rot) | (xs << ((32 - rot) & 31))) & M
        out += struct.pack("<I", x)
    return out


def make_rng():
    kat = {
        "chacha20_zero_key_block0_words": [0xADE0B876, 0x903DF1A0, 0xE56A5D40, 0x28BD8653],
        "rand085_test_stdrng_construction": {
            "seed": [1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16,
            "x0_next_u64": 10719222850664546238,
            "x1_next_u64_after_from_rng": 14064965282130556830,
        },
    }
    sd = seed_from_u64(42)
    r = PyStdRng(sd)
    kat["seed_from_u64_42"] = {"seed_hex": sd.hex(), "first8_u32": [r.u32() for _ in range(8)]}
    # index streams: base.rs:384-390 for three (size, batch) settings; first batch verbatim and a
    # sha256 over the first 1000 batches (little-endian u64 each).
    streams = []
    for seed, size, batch in [(42, 1_000_000, 256), (42, 10_000, 32), (7, 65_537, 512), (42, 1, 4)]:
        r = PyStdRng(seed_from_u64(seed))
        h = hashlib.sha256()
        first = None
        for b in range(1000):
            ixs = [r.u32() % size for _ in range(batch)]
            if b == 0:
                first = ixs
            h.update(struct.pack("<%dQ" % batch, *ixs))
        streams.append(dict(seed=seed, size=size, batch=batch, first_batch=first, sha256_1000_batches=h.hexdigest()))
    kat["index_streams"] = streams
    with open(os.path.join(HERE, "rng_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)


def sample_stride(n):
    return max(1, n // 4096) | 1


def per_weights(s, n):
    """importance weights of the PER fixtures (fixed-seed uniform in (0.2, 1])"""
    return (0.2 + 0.8 * np.random.default_rng(900 + s).random(n)).astype(np.float32)


def make_dqn(name, kind, shapes, batch_fn, n_steps, weighted=False, **kw):
    from oracle import torch_ref as T
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(1)
    p0 = T.init_params(shapes, seed=kw.pop("param_seed"))
    agent = T.TorchDqn(kind, shapes, p0, **kw)
    out = {}
    for s in range(n_steps):
        obs, act, nobs, rew, term = batch_fn(s)
        r = agent.update(obs, act, nobs, rew, term, weight=per_weights(s, len(rew)) if weighted else None)
        if weighted:
            out[f"s{s}_td_errs"] = r["td_abs"]
        out[f"s{s}_loss"] = np.float32(r["loss"])
        out[f"s{s}_q_pred_all"] = r["q_pred_all"]
        out[f"s{s}_q_next_all"] = r["q_next_all"]
        out[f"s{s}_pred"], out[f"s{s}_tgt"] = r["pred"], r["tgt"]
        st = sample_stride(r["grads"].size)
        out[f"s{s}_grads_sample"] = r["grads"][::st]
        out[f"s{s}_params_sample"] = agent.params()[::st]
        out[f"s{s}_tgt_params_sample"] = agent.tgt_params()[::st]
        # per-variable L2 norms
        o, gn, pn = 0, [], []
        P = agent.params()
        for sh in shapes:
            n = int(np.prod(sh))
            gn.append(np.linalg.norm(r["grads"][o:o + n].astype(np.float64)))
            pn.append(np.linalg.norm(P[o:o + n].astype(np.float64)))
            o += n
        out[f"s{s}_grad_norms"], out[f"s{s}_param_norms"] = np.array(gn), np.array(pn)
    np.savez_compressed(os.path.join(HERE, name), **out)


SAC_CASES = {
    # name: (obs_dim, act_dim, pi_units, q_units, n_critics, B, steps, kwargs)
    "sac_17_6_twinq_auto": (17, 6, [64, 64], [64, 64], 2, 32, 3,
                            dict(lr_actor=3e-4, lr_critic=3e-4, ent_coef=("Auto", -6.0, 3e-4), critic_loss="Mse")),
    "sac_pendulum_fix_huber": (3, 1, [64, 64], [64, 64], 1, 16, 2,
                               dict(lr_actor=3e-4, lr_critic=3e-4, ent_coef=("Fix", 1.0), critic_loss="SmoothL1",
                                    reward_scale=0.5)),
    # OptimizerConfig::AdamW on both models (opt.rs:20-27, 38-55): the actor without, the twin critics with amsgrad
    "sac_17_6_twinq_adamw": (17, 6, [64, 64], [64, 64], 2, 32, 4,
                             dict(lr_actor=1e-3, lr_critic=2e-3, ent_coef=("Auto", -6.0, 3e-4), critic_loss="Mse",
                                  adamw_actor=dict(beta1=0.85, beta2=0.97, wd=0.02, eps=1e-6, amsgrad=False),
                                  adamw_critic=dict(beta1=0.8, beta2=0.9, wd=0.05, eps=1e-6, amsgrad=True))),
}


def sac_case_params(name):
    from oracle import torch_ref as T
    od, ad, pu, qu, nc, B, steps, kw = SAC_CASES[name]
    seed = sum(map(ord, name))
    pi0 = T.init_params(T.sac_pi_shapes(od, pu, ad), seed) * np.float32(0.5)
    q0 = [T.init_params(T.sac_q_shapes(od, ad, qu), seed + 1 + i) for i in range(nc)]
    return od, ad, pu, qu, nc, B, steps, kw, pi0, q0, seed


def sac_case_batch(name, s):
    """minibatch + noise of step s; the AdamW case scales rewards 10x on its first two steps and 0.1x after, so that exp_avg_sq
    decays below its running maximum and amsgrad is not a no-op"""
    from oracle import torch_ref as T
    od, ad, pu, qu, nc, B, steps, kw = SAC_CASES[name]
    obs, act, nobs, rew, term, za, zn = T.sac_batch(B, od, ad, sum(map(ord, name)) + 100 + s)
    if "adamw_critic" in kw:
        rew = (rew * (10.0 if s < 2 else 0.1)).astype(np.float32)
    return obs, act, nobs, rew, term, za, zn


def make_sac():
    """SAC goldens: PyTorch CPU autograd on seeded minibatches with injected N(0,1) noise."""
    from oracle import torch_ref as T
    import torch
    torch.set_num_threads(1)
    for name in SAC_CASES:
        od, ad, pu, qu, nc, B, steps, kw, pi0, q0, seed = sac_case_params(name)
        agent = T.TorchSac(od, ad, pu, qu, pi0, q0, **kw)
        out = {}
        for s in range(steps):
            r = agent.update(*sac_case_batch(name, s))
            for k in ("loss_critic", "loss_actor", "ent_coef", "log_alpha"):
                out[f"s{s}_{k}"] = np.float32(r[k])
            out[f"s{s}_a"], out[f"s{s}_log_p"], out[f"s{s}_tgt"] = r["a"], r["log_p"], r["tgt"]
            out[f"s{s}_pi_grads"], out[f"s{s}_pi_params"] = r["pi_grads"], r["pi_params"]
            for i in range(nc):
                out[f"s{s}_q{i}_grads"], out[f"s{s}_q{i}_params"] = r["q_grads"][i], r["q_params"][i]
                out[f"s{s}_q{i}_tgt_params"] = r["q_tgt_params"][i]
        if "adamw_critic" in kw:   # the second moment and its running maximum after the last step
            for i in range(nc):
                out[f"q{i}_exp_avg_sq"] = T.flatten(agent.opt[f"q{i}"]["v"])
                out[f"q{i}_max_exp_avg_sq"] = T.flatten(agent.opt[f"q{i}"]["vmax"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


IQN_CASES = {
    # name: (psi_kind, feature_dim, embed_dim, f_units, n_actions, psi_in, psi_units, B, n_pred, n_tgt, steps, lr)
    "iqn_mlp_small": ("mlp", 64, 16, [32], 3, 5, [48], 8, 8, 8, 3, 1e-3),
    "iqn_cnn_b2": ("cnn", 3136, 64, [512], 6, None, [], 2, 8, 8, 2, 1e-4),
    "iqn_mlp_small_adamw": ("mlp", 64, 16, [32], 3, 5, [48], 8, 8, 8, 4, 2e-3),
}
# IqnModelConfig.opt_config = OptimizerConfig::AdamW (opt.rs:20-27, 38-55) of the cases that have one
IQN_ADAMW = {"iqn_mlp_small_adamw": dict(beta1=0.8, beta2=0.9, wd=0.05, eps=1e-6, amsgrad=True)}


def iqn_case_batch(name, s):
    """minibatch + percent points of step s (the AdamW case scales rewards so that amsgrad is not a no-op, as sac_case_batch)"""
    from oracle import torch_ref as T
    kind, F_, E, fu, A, pin, pu, B, n_p, n_t, steps, lr = IQN_CASES[name]
    b = list(T.iqn_batch(B, kind, A, n_p, n_t, sum(map(ord, name)) + 50 + s, in_dim=pin))
    if name in IQN_ADAMW:
        b[3] = (b[3] * (10.0 if s < 2 else 0.1)).astype(np.float32)
    return tuple(b)


def iqn_case(name):
    from oracle import torch_ref as T
    kind, F_, E, fu, A, pin, pu, B, n_p, n_t, steps, lr = IQN_CASES[name]
    sh = T.iqn_shapes(kind, F_, E, fu, A, psi_in=pin, psi_units=pu)
    seed = sum(map(ord, name))
    p0 = T.init_params(sh[0] + sh[1] + sh[2], seed)
    return kind, F_, E, fu, A, pin, pu, B, n_p, n_t, steps, lr, sh, p0, seed


def make_iqn():
    """IQN goldens: PyTorch CPU autograd, injected percent points; CNN case stores strided samples."""
    from oracle import torch_ref as T
    import torch
    torch.set_num_threads(1)
    for name in IQN_CASES:
        kind, F_, E, fu, A, pin, pu, B, n_p, n_t, steps, lr, sh, p0, seed = iqn_case(name)
        agent = T.TorchIqn(kind, sh, p0, lr=lr, feature_dim=F_, embed_dim=E, tau=0.01, soft_update_interval=2, adamw=IQN_ADAMW.get(name))
        st = sample_stride(p0.size)
        out = {}
        for s in range(steps):
            r = agent.update(*iqn_case_batch(name, s))
            out[f"s{s}_loss"] = np.float32(r["loss"])
            out[f"s{s}_z_pred"], out[f"s{s}_z_tgt"], out[f"s{s}_tgt"] = r["z_pred"], r["z_tgt"], r["tgt"]
            out[f"s{s}_grads_sample"], out[f"s{s}_params_sample"] = r["grads"][::st], r["params"][::st]
            out[f"s{s}_tgt_params_sample"] = r["tgt_params"][::st]
            o, gn = 0, []
            for shp in sh[0] + sh[1] + sh[2]:
                n = int(np.prod(shp))
                gn.append(np.linalg.norm(r["grads"][o:o + n].astype(np.float64)))
                o += n
            out[f"s{s}_grad_norms"] = np.array(gn)
        if name in IQN_ADAMW:
            out["exp_avg_sq"], out["max_exp_avg_sq"] = T.flatten(agent.v), T.flatten(agent.vmax)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def main():
    make_rng()
    from oracle import torch_ref as T

    # (1) Nature-CNN, B=4, A=6, SmoothL1, 3 steps, tau=1 sync every 2 steps
    make_dqn("dqn_cnn_b4_huber.npz", "cnn", T.cnn_shapes(6), lambda s: T.synthetic_atari_batch(4, 6, 100 + s), 3,
             param_seed=1, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=2)
    # (2) Nature-CNN, B=8, A=6, Mse + double DQN (the Atari example's loss), 2 steps
    make_dqn("dqn_cnn_b8_mse_ddqn.npz", "cnn", T.cnn_shapes(6), lambda s: T.synthetic_atari_batch(8, 6, 200 + s), 2,
             param_seed=2, lr=1e-4, critic_loss="Mse", double_dqn=True, tau=0.005, soft_update_interval=1)

    # (3) CartPole-shaped MLP[64,64], B=32 (BASELINE config 1), 5 steps, tau=.01 every step
    def cart(s):
        rng = np.random.default_rng(300 + s)
        obs = rng.standard_normal((32, 4)).astype(np.float32)
        nobs = rng.standard_normal((32, 4)).astype(np.float32)
        act = rng.integers(0, 2, 32)
        rew = np.ones(32, np.float32)
        term = (rng.random(32) < 0.1).astype(np.int8)
        return obs, act, nobs, rew, term

    make_dqn("dqn_mlp_cartpole.npz", "mlp", T.mlp_shapes(4, [64, 64], 2), cart, 5,
             param_seed=3, lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1)
    # (4) the importance-weighted branch of update_critic (dqn/base.rs:123-145): SmoothL1 without clipping,
    #     Mse with clip_td_err; rewards scaled so that |td| straddles both the Huber knee and the clip range
    def cart_per(s):
        obs, act, nobs, rew, term = cart(s)
        return obs, act, nobs, (rew * np.random.default_rng(700 + s).uniform(-2, 2, 32)).astype(np.float32), term

    make_dqn("dqn_mlp_per_huber.npz", "mlp", T.mlp_shapes(4, [64, 64], 2), cart_per, 3, weighted=True,
             param_seed=4, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1)
    make_dqn("dqn_mlp_per_mse_clip.npz", "mlp", T.mlp_shapes(4, [64, 64], 2), cart_per, 3, weighted=True,
             param_seed=5, lr=1e-3, critic_loss="Mse", clip_td_err=(0.05, 0.9), double_dqn=True, tau=0.01, soft_update_interval=1)
    make_sac()
    make_iqn()
    print("fixtures:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
