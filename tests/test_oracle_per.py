"""The PER restatement against the reference's own known-answer test
(border-core/src/generic_replay_buffer/base/sum_tree.rs:180-217, `test_sum_tree_odd`) and the
IwScheduler formula (base/iw_scheduler.rs:35-43)."""
import numpy as np

from oracle.oracle import PerReplay, SumTree, iw_beta

DATA = [0.5, 0.2, 0.8, 0.3, 1.1, 2.5, 3.9]      # sum_tree.rs:184


def _tree():
    t = SumTree(8, 1.0, "Batch")                # :185
    for ix, p in enumerate(DATA):
        t.add(ix, p)
    return t


def test_reference_get_kats():
    t = _tree()
    # sum_tree.rs:192-199
    for s, want in [(0.0, 0), (0.4, 0), (0.5, 0), (0.6, 1), (1.2, 2), (1.6, 3), (2.0, 4), (2.8, 4)]:
        assert t.get(s) == want, (s, t.get(s), want)
    assert t.n_samples == 7
    assert abs(t.total() - sum(DATA)) < 1e-5
    t.update(7, 2.0)                             # :201
    assert abs(t.total() - (sum(DATA) + 2.0)) < 1e-5
    assert t.get(t.total()) == 7                 # the new leaf owns the top of the range


def test_tree_is_consistent_and_max_min():
    t = _tree()
    tr = t.tree()
    cap = 8
    for i in range(cap - 1):                      # every internal node == sum of its children (up to f32 drift)
        assert abs(tr[i] - (tr[2 * i + 1] + tr[2 * i + 2])) < 1e-5
    np.testing.assert_allclose(tr[cap - 1:cap - 1 + 7], np.float32(DATA) + np.float32(1e-8), rtol=1e-7)
    assert abs(t.max() - 3.9) < 1e-6 and abs(t.min_p() - 0.2) < 1e-6


def test_sample_weights_formula():
    t = SumTree(8, 1.0, "All")
    for ix, p in enumerate(DATA):
        t.add(ix, p)
    u = np.linspace(0.01, 0.99, 64).astype(np.float32)
    ixs, ws = t.sample(u, 0.5)
    tot = np.float32(t.total())
    leaf = (np.float32(DATA) + np.float32(1e-8))[ixs]
    n = np.float32(7) / tot
    want = (n * leaf) ** np.float32(-0.5) * (n * np.float32(0.2 + 1e-8)) ** np.float32(0.5)
    np.testing.assert_allclose(ws, want, rtol=2e-6)
    assert ws.max() <= 1.0 + 1e-6                  # All: normalised by the largest possible weight
    # indices follow the priorities: empirical frequencies ~ p / total
    ixs2, ws2 = t.sample(np.random.default_rng(0).random(20000).astype(np.float32), 0.5)
    freq = np.bincount(ixs2, minlength=8)[:7] / 20000.0
    assert np.abs(freq - np.float32(DATA) / tot).max() < 0.01
    tb = SumTree(8, 1.0, "Batch")
    for ix, p in enumerate(DATA):
        tb.add(ix, p)
    _, wb = tb.sample(u, 0.5)
    assert abs(wb.max() - 1.0) < 1e-6              # Batch: the batch's own maximum is 1


def test_iw_scheduler_beta():
    assert iw_beta(0.4, 1.0, 500_000, 0) == np.float32(0.4)
    assert abs(iw_beta(0.4, 1.0, 500_000, 250_000) - 0.7) < 1e-6
    assert iw_beta(0.4, 1.0, 500_000, 500_000) == 1.0 and iw_beta(0.4, 1.0, 500_000, 10**9) == 1.0


def test_per_replay_push_sets_max_priority_and_update_changes_sampling():
    r = PerReplay(64, 42, alpha=0.6, normalize="All")
    r.push(10)
    # all new rows carry the same (initial max) priority -> uniform over the 10 rows
    ixs, ws = r.batch(2000)
    assert ixs.min() >= 0 and ixs.max() < 10
    np.testing.assert_allclose(ws, 1.0, rtol=1e-5)
    r.update_priority(np.arange(10), np.array([10.0] + [0.01] * 9, np.float32))
    assert r.n_opts == 1
    ixs, ws = r.batch(4000)
    assert (ixs == 0).mean() > 0.8                 # row 0 dominates
    assert ws[ixs == 0].max() < ws[ixs != 0].min() # and carries the smallest importance weight
    r.push(3)                                      # new rows enter at the current maximum priority
    ixs, _ = r.batch(4000)
    assert 0.2 < np.isin(ixs, [10, 11, 12]).mean() < 0.95
