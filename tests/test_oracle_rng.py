"""The C oracle's StdRng / index sampling against the committed known-answer vectors
(tests/golden/rng_kat.json: rand 0.8.5's own KAT + an independent pure-Python ChaCha12)."""
import hashlib
import json
import os
import struct

import numpy as np

from oracle import oracle as O


def _kat(golden_dir):
    with open(os.path.join(golden_dir, "rng_kat.json")) as f:
        return json.load(f)


def test_chacha20_zero_key_vector(golden_dir):
    k = _kat(golden_dir)
    blk = O.chacha_block([0] * 8, 0, 20)
    assert [int(x) for x in blk[:4]] == k["chacha20_zero_key_block0_words"]


def test_rand085_stdrng_construction_kat(golden_dir):
    """rand-0.8.5/src/rngs/std.rs test_stdrng_construction: value stability of StdRng."""
    k = _kat(golden_dir)["rand085_test_stdrng_construction"]
    r0 = O.StdRng.from_seed(bytes(k["seed"]))
    assert r0.next_u64() == k["x0_next_u64"]
    # StdRng::from_rng(rng0): fill 32 seed bytes from rng0 (8 little-endian words)
    seed = b"".join(struct.pack("<I", r0.next_u32()) for _ in range(8))
    r1 = O.StdRng.from_seed(seed)
    assert r1.next_u64() == k["x1_next_u64_after_from_rng"]


def test_seed_from_u64(golden_dir):
    k = _kat(golden_dir)["seed_from_u64_42"]
    assert O.seed_bytes_from_u64(42).hex() == k["seed_hex"]
    r = O.StdRng.seed_from_u64(42)
    assert [r.next_u32() for _ in range(8)] == k["first8_u32"]


def test_index_streams(golden_dir):
    """generic_replay_buffer/base.rs:384-390 -- first 1000 batches, bit-exact."""
    for s in _kat(golden_dir)["index_streams"]:
        r = O.StdRng.seed_from_u64(s["seed"])
        h = hashlib.sha256()
        for b in range(1000):
            ixs = r.sample_indices(s["size"], s["batch"])
            if b == 0:
                assert ixs.tolist() == s["first_batch"]
            h.update(ixs.astype("<u8").tobytes())
        assert h.hexdigest() == s["sha256_1000_batches"]


def test_modulo_bias_is_kept():
    """`next_u32() % size` (with replacement, modulo-biased) -- not a rejection sampler."""
    r = O.StdRng.seed_from_u64(1)
    w = O.StdRng.seed_from_u64(1)
    size = 3_000_000_000
    ixs = r.sample_indices(size, 64)
    raw = np.array([w.next_u32() for _ in range(64)], dtype=np.uint64)
    assert (ixs == raw % size).all()


def test_xoshiro256pp_known_answers_and_lane_seeding():
    """The optional device-native index generator (bdr_replay_config::index_rng = 1; not the reference's stream): xoshiro256++ 1.0's
    published first outputs from state {1, 2, 3, 4}, SplitMix64's published first outputs from seed 1234567, and the lane layout."""
    g = O.Xoshiro256pp([1, 2, 3, 4])
    assert [g.next_u64() for _ in range(5)] == [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205]
    assert [O.splitmix64_at(1234567, i) for i in range(5)] == [6457827717110365317, 3203168211198807973, 9817491932198370423,
                                                                  4593380528125082431, 16408922859458223821]
    a, b = O.XoshiroLanes(7), O.XoshiroLanes(7)
    first = a.sample_indices(1000, 16)
    assert (first == b.sample_indices(1000, 32)[:16]).all()          # lane j is the same generator whatever the batch size
    assert (first < 1000).all() and len(set(first.tolist())) > 8
    second = a.sample_indices(1000, 16)
    assert (second != first).any()
    lane3 = O.Xoshiro256pp([O.splitmix64_at(7, 12 + i) for i in range(4)])
    assert first[3] == (lane3.next_u64() >> 32) % 1000 and second[3] == (lane3.next_u64() >> 32) % 1000
