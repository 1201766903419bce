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
