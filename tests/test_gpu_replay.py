"""HIP replay path vs the CPU oracle (through the C ABI): bit-exact indices and gathered bytes."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def _tr(rng, n, obs_shape, n_actions=6):
    return (rng.integers(0, 256, (n,) + obs_shape, dtype=np.uint8), rng.integers(0, n_actions, (n, 1)).astype(np.int64),
            rng.integers(0, 256, (n,) + obs_shape, dtype=np.uint8), rng.standard_normal(n).astype(np.float32),
            (rng.random(n) < .3).astype(np.int8), (rng.random(n) < .3).astype(np.int8))


def test_index_streams_match_golden(B, golden_dir):
    """First 1000 index batches, bit-identical to the committed StdRng vectors (base.rs:384-390)."""
    kat = json.load(open(os.path.join(golden_dir, "rng_kat.json")))
    for s in kat["index_streams"]:
        cap = s["size"]
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=s["seed"]), (4,), np.float32)
        rb.fill_synthetic(cap, seed=1, kind=1, n_actions=3)
        h = hashlib.sha256()
        for b in range(1000):
            ixs = rb.sample_indices(s["batch"])
            if b == 0:
                assert ixs.tolist() == s["first_batch"]
            h.update(ixs.astype("<u8").tobytes())
        assert h.hexdigest() == s["sha256_1000_batches"], s
        rb.close()


def test_push_batch_matches_oracle_ragged(B):
    """Ragged pushes with wrap-around, then batches: every field bit-identical to the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    cap, obs_shape = 37, (4, 1, 84, 84)
    ob = int(np.prod(obs_shape))
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=9), obs_shape, np.uint8)
    ref = O.Replay(cap, 9, ob, 8)
    for n in [1, 5, 30, 7, 1, 36, 2]:
        t = _tr(rng, n, obs_shape)
        rb.push(*t)
        ref.push(*t)
        assert len(rb) == len(ref) and rb.head == ref.head
        for bs in (1, 16):
            g, r = rb.batch(bs), ref.batch(bs)
            assert (g.ix_sample == r["ixs"]).all()
            assert (g.obs.reshape(bs, -1) == r["obs"]).all() and (g.next_obs.reshape(bs, -1) == r["next_obs"]).all()
            assert (g.act.view(np.uint8).reshape(bs, -1) == r["act"]).all()
            assert (g.reward == r["reward"]).all()
            assert (g.is_terminated == r["is_terminated"]).all() and (g.is_truncated == r["is_truncated"]).all()
    rb.close()


def test_small_f32_rows(B):
    """CartPole / SAC shaped rows (f32 obs not a multiple of 16 bytes, f32 actions)."""
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100, seed=3), (17,), np.float32, (6,), np.float32)
    ref = O.Replay(100, 3, 68, 24)
    for n in [10, 95, 13]:
        t = (rng.standard_normal((n, 17)).astype(np.float32), rng.uniform(-1, 1, (n, 6)).astype(np.float32),
             rng.standard_normal((n, 17)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
             np.zeros(n, np.int8), np.zeros(n, np.int8))
        rb.push(*t)
        ref.push(*t)
    g, r = rb.batch(64), ref.batch(64)
    assert (g.ix_sample == r["ixs"]).all()
    assert (g.obs.view(np.uint8).reshape(64, -1) == r["obs"]).all()
    assert (g.act.view(np.uint8).reshape(64, -1) == r["act"]).all()
    rb.close()


def test_empty_buffer_errors(B):
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=8), (4,), np.float32)
    with pytest.raises(B.BdrError) as e:
        rb.batch(2)
    assert e.value.code == 4
    rb.close()


def test_synthetic_fill_matches_host_restatement(B):
    from tests import synth
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=64, seed=42), (4, 1, 84, 84), np.uint8)
    rb.fill_synthetic(64, seed=5, kind=0, n_actions=6)
    obs, act, nobs, rew, term, trunc = rb.read_rows(3, 20)
    e = synth.atari_rows(5, 3, 20)
    assert (obs.reshape(20, -1) == e[0]).all() and (nobs.reshape(20, -1) == e[2]).all()
    assert (act[:, 0] == e[1]).all() and (rew == e[3]).all() and (term == e[4]).all() and (trunc == 0).all()
    rb.close()


def test_full_size_gather_roundtrip(B):
    """BASELINE-sized rows at batch 256 on a device-filled ring: gathered rows == ring rows at the
    sampled indices (size-independent property; ring kept small enough for a quick test)."""
    from oracle import oracle as O
    from tests import synth
    cap = 20000
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4, 1, 84, 84), np.uint8)
    rb.fill_synthetic(cap, seed=0, kind=0, n_actions=6)
    ref = O.StdRng.seed_from_u64(42)
    for _ in range(3):
        g = rb.batch(256)
        ixs = ref.sample_indices(cap, 256)
        assert (g.ix_sample == ixs).all()
        for k in (0, 100, 255):
            e = synth.atari_rows(0, int(ixs[k]), 1)
            assert (g.obs[k].ravel() == e[0][0]).all() and (g.next_obs[k].ravel() == e[2][0]).all()
            assert g.act[k, 0] == e[1][0] and g.reward[k] == e[3][0] and g.is_terminated[k] == e[4][0]
    rb.close()


def test_baseline_size_ring_one_million_transitions(B):
    """BASELINE configuration 2 at its real size: a 1 000 000-transition ring of [4,1,84,84] u8 rows (2 x 28.2 GB of HBM),
    batch 256.  SURVEY 8(d): the indices of the first 1 000 batches are identical to the CPU restatement of
    `ReplayBufferBase::batch` (base.rs:384-390); the gathered rows are the ring rows at those indices (checked against the
    counter-based fill, which is a function of the transition index alone, and by a checksum of checksums over whole
    batches); pushes at the end of the ring wrap to row 0 (base.rs:295-316) and are what batch() returns afterwards."""
    import ctypes as C
    from border_amd import _lib
    from oracle import oracle as O
    from tests import synth
    cap, bs = 1_000_000, 256
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4, 1, 84, 84), np.uint8)
    rb.fill_synthetic(cap - 2, seed=0, kind=0, n_actions=6)
    assert rb.len() == cap - 2 and rb.head == cap - 2
    # wrap: 5 rows pushed at cap-2 land in rows cap-2, cap-1, 0, 1, 2
    rng = np.random.default_rng(0)
    pobs = rng.integers(0, 256, (5, 4, 1, 84, 84), dtype=np.uint8)
    pnext = rng.integers(0, 256, (5, 4, 1, 84, 84), dtype=np.uint8)
    pact = np.arange(5, dtype=np.int64).reshape(5, 1)
    prew = np.array([0.5, -1.5, 2.5, 3.5, -4.5], np.float32)
    pterm = np.array([1, 0, 0, 1, 0], np.int8)
    rb.push(pobs, pact, pnext, prew, pterm, np.zeros(5, np.int8))
    assert rb.len() == cap and rb.head == 3
    for row, k in ((cap - 2, 0), (cap - 1, 1), (0, 2), (1, 3), (2, 4)):
        o, a, n, r, t, _ = rb.read_rows(row, 1)
        assert (o[0] == pobs[k]).all() and (n[0] == pnext[k]).all() and a[0, 0] == k and r[0] == prew[k] and t[0] == pterm[k]
    pushed = {cap - 2: 0, cap - 1: 1, 0: 2, 1: 3, 2: 4}

    def row_sums(ixs):   # per-row byte sums of (obs, next_obs) from the definition of the ring contents
        so, sn = np.empty(len(ixs), np.int64), np.empty(len(ixs), np.int64)
        for j, ix in enumerate(ixs):
            ix = int(ix)
            if ix in pushed:
                so[j], sn[j] = pobs[pushed[ix]].sum(dtype=np.int64), pnext[pushed[ix]].sum(dtype=np.int64)
            else:
                e = synth.atari_rows(0, ix, 1)
                so[j], sn[j] = e[0].sum(dtype=np.int64), e[2].sum(dtype=np.int64)
        return so, sn

    ref = O.StdRng.seed_from_u64(42)
    ixs = np.empty(bs, np.uint64)
    L = _lib.lib()
    for it in range(1000):
        want = ref.sample_indices(cap, bs)
        if it % 250 == 0:    # full host copy: exact rows, and the checksum of checksums over the whole batch
            g = rb.batch(bs)
            assert (g.ix_sample == want).all()
            so, sn = row_sums(want)
            assert (g.obs.reshape(bs, -1).sum(1, dtype=np.int64) == so).all()
            assert (g.next_obs.reshape(bs, -1).sum(1, dtype=np.int64) == sn).all()
            assert int(g.obs.sum(dtype=np.int64)) == int(so.sum()) and int(g.next_obs.sum(dtype=np.int64)) == int(sn.sum())
            k = int(np.argmax(want))
            if int(want[k]) not in pushed:
                e = synth.atari_rows(0, int(want[k]), 1)
                assert (g.obs[k].ravel() == e[0][0]).all() and (g.next_obs[k].ravel() == e[2][0]).all()
                assert g.act[k, 0] == e[1][0] and g.reward[k] == e[3][0] and g.is_terminated[k] == e[4][0]
        else:                # device-side batch, indices only
            _lib.check(L.bdr_replay_batch(rb.handle, bs, ixs.ctypes.data_as(C.c_void_p), None, None, None, None, None, None))
            assert (ixs == want).all(), it
    assert int(want.max()) < cap
    rb.close()


def test_sac_baseline_ring_one_million_f32_transitions(B):
    """BASELINE configuration 5 at its real size: a 1 000 000-transition ring of HalfCheetah-shaped rows (obs 17 x f32,
    act 6 x f32), batch 1024.  The indices of the first 1 000 batches are identical to the CPU restatement of
    `ReplayBufferBase::batch` (base.rs:384-390); the gathered rows are the ring rows at those indices (every field of every
    row against the counter-based fill, a function of the transition index alone); pushes at the end of the ring wrap to
    row 0 (base.rs:295-316)."""
    import ctypes as C
    from border_amd import _lib
    from oracle import oracle as O
    from tests import synth
    cap, bs, od, ad = 1_000_000, 1024, 17, 6
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (od,), np.float32, (ad,), np.float32)
    rb.fill_synthetic(cap - 3, seed=7, kind=1, n_actions=0)
    assert rb.len() == cap - 3 and rb.head == cap - 3
    rng = np.random.default_rng(1)
    pobs = rng.standard_normal((8, od)).astype(np.float32)
    pnext = rng.standard_normal((8, od)).astype(np.float32)
    pact = rng.uniform(-1, 1, (8, ad)).astype(np.float32)
    prew = rng.standard_normal(8).astype(np.float32)
    pterm = (rng.random(8) < .5).astype(np.int8)
    rb.push(pobs, pact, pnext, prew, pterm, np.zeros(8, np.int8))
    assert rb.len() == cap and rb.head == 5
    pushed = {(cap - 3 + k) % cap: k for k in range(8)}

    def rows(ixs):
        ixs = np.asarray(ixs, np.uint64)
        obs, act, nobs, rew, term = (np.empty((len(ixs), od), np.float32), np.empty((len(ixs), ad), np.float32),
                                     np.empty((len(ixs), od), np.float32), np.empty(len(ixs), np.float32), np.empty(len(ixs), np.int8))
        t = ixs[:, None]
        obs[:] = synth._normal(synth.synth_hash(7, t, 0, np.arange(od, dtype=np.uint64)[None]))
        nobs[:] = synth._normal(synth.synth_hash(7, t, 1, np.arange(od, dtype=np.uint64)[None]))
        act[:] = (synth.synth_hash(7, t, 3, np.arange(ad, dtype=np.uint64)[None]) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
        rew[:] = synth._normal(synth.synth_hash(7, ixs, 2, 2))
        term[:] = ((synth.synth_hash(7, ixs, 2, 1) >> np.uint64(40)).astype(np.int64) < 83886)
        for j, ix in enumerate(ixs):
            k = pushed.get(int(ix))
            if k is not None:
                obs[j], act[j], nobs[j], rew[j], term[j] = pobs[k], pact[k], pnext[k], prew[k], pterm[k]
        return obs, act, nobs, rew, term

    e = synth.f32_rows(7, 100, 4, od, ad)       # the vectorised helper and the row-wise form agree
    r = rows(np.arange(100, 104))
    assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(e[:5], r))
    ref = O.StdRng.seed_from_u64(42)
    ixs = np.empty(bs, np.uint64)
    L = _lib.lib()
    for it in range(1000):
        want = ref.sample_indices(cap, bs)
        if it % 100 == 0:
            g = rb.batch(bs)
            assert (g.ix_sample == want).all()
            o, a, n, rw, tm = rows(want)
            assert (g.obs.view(np.uint32) == o.view(np.uint32)).all() and (g.next_obs.view(np.uint32) == n.view(np.uint32)).all()
            assert (g.act.view(np.uint32) == a.view(np.uint32)).all() and (g.reward.view(np.uint32) == rw.view(np.uint32)).all()
            assert (g.is_terminated == tm).all() and (g.is_truncated == 0).all()
        else:
            _lib.check(L.bdr_replay_batch(rb.handle, bs, ixs.ctypes.data_as(C.c_void_p), None, None, None, None, None, None))
            assert (ixs == want).all(), it
    # the pushed rows are what the ring holds at the wrap
    o, a, n, rw, tm, _ = rb.read_rows(cap - 3, 3)
    assert (o == pobs[:3]).all() and (a == pact[:3]).all() and (rw == prew[:3]).all() and (tm == pterm[:3]).all()
    o, a, n, rw, tm, _ = rb.read_rows(0, 5)
    assert (o == pobs[3:]).all() and (n == pnext[3:]).all() and (rw == prew[3:]).all()
    rb.close()


def test_xoshiro_index_generator_matches_its_restatement(B):
    """bdr_replay_config::index_rng = BDR_RNG_XOSHIRO256PP: one xoshiro256++ generator per batch lane in HBM.  Not the reference's
    StdRng stream (that is the default, pinned above) - checked bit for bit against oracle.XoshiroLanes, ragged batch sizes included,
    and the gathered rows are the rows those indices name."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    cap, obs_shape = 1000, (6,)
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=77, index_rng="xoshiro256++"), obs_shape, np.float32)
    obs = rng.standard_normal((700,) + obs_shape).astype(np.float32)
    nxt = rng.standard_normal((700,) + obs_shape).astype(np.float32)
    act = rng.integers(0, 3, (700, 1)).astype(np.int64)
    rew = rng.standard_normal(700).astype(np.float32)
    z = np.zeros(700, np.int8)
    rb.push(obs, act, nxt, rew, z, z)
    ref = O.XoshiroLanes(77)
    for bs in (1, 64, 257, 32, 4096, 3):
        want = ref.sample_indices(700, bs)
        if bs % 2:
            got = rb.sample_indices(bs)
        else:
            g = rb.batch(bs)
            got = g.ix_sample
            assert (g.obs.reshape(bs, -1) == obs[want.astype(np.int64)]).all() and (g.reward == rew[want.astype(np.int64)]).all()
            assert (g.next_obs.reshape(bs, -1) == nxt[want.astype(np.int64)]).all() and (g.act.reshape(-1) == act[want.astype(np.int64), 0]).all()
        assert (got == want).all(), bs
    with pytest.raises(B.BdrError, match="at most"):
        rb.sample_indices((1 << 16) + 1)
    rb.close()
    # prioritized sampling draws from the StdRng stream: the combination is refused, not silently ignored
    with pytest.raises(B.BdrError, match="StdRng"):
        B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=64, seed=1, index_rng="xoshiro256++", per_config=B.PerConfig()), (4,), np.float32)

