"""tools/upstream_kat (the Cargo project that produces the three upstream pins on a machine with cargo) and its ingest path
(tests/golden/ingest_upstream.py), exercised end to end against CANDIDATE output generated from the oracle itself - so the day
real output arrives, one command folds it in.  When tests/golden/upstream_pins.json exists (someone ran it), the oracle is held to
the pinned values here, on every CPU run."""
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = os.path.join(ROOT, "tools", "upstream_kat")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ingest_upstream as I  # noqa: E402


@pytest.fixture(scope="module")
def candidate(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("kat") / "out")
    subprocess.check_call([sys.executable, os.path.join(KAT, "candidate.py"), out], cwd=ROOT)
    return out


def test_cargo_project_pins_the_crates_border_pins():
    toml = open(os.path.join(KAT, "Cargo.toml")).read()
    assert re.search(r'rand\s*=\s*\{\s*version\s*=\s*"=0\.8\.5"', toml) and 'tch = "0.16.0"' in toml and 'image = "0.23.14"' in toml
    ref = "/root/reference/Cargo.toml"
    if os.path.exists(ref):   # (this container only)
        r = open(ref).read()
        assert 'rand = { version = "=0.8.5"' in r and 'tch = "0.16.0"' in r and 'image = "0.23.14"' in r
    rs = open(os.path.join(KAT, "src", "main.rs")).read()
    for needle in ("StdRng::seed_from_u64", "next_u32", "next_u64", "fill_bytes", "VarStore::new", "vs.save(", "resize(&img, 84, 84, Triangle)",
                   "upstream_kat.json"):
        assert needle in rs, needle
    # the two frame generators use the same constants
    py = open(os.path.join(ROOT, "tests", "golden", "ingest_upstream.py")).read()
    for const in ("1664525", "1013904223", "12345", "[200, 72, 72]", "[84, 138, 210]", "% 37", "% 29", "== 100", "== 77", "== 151"):
        assert const in rs and const in py, const
    assert [int(x, 0) for x in re.search(r"const SEEDS: \[u64; 6\] = \[(.*?)\];", rs).group(1).replace("_", "").split(",")] == I.SEEDS


def test_ingest_accepts_the_candidate_values(candidate):
    report, doc = I.check(candidate)
    assert doc["producer"] == "candidate"
    assert len(report) >= 12
    assert all(ok for _, ok, _ in report), [r for r in report if not r[1]]


def test_ingest_names_the_first_mismatch_and_writes_nothing(candidate, tmp_path):
    bad = str(tmp_path / "bad")
    shutil.copytree(candidate, bad)
    doc = json.load(open(os.path.join(bad, "upstream_kat.json")))
    doc["rng"][0]["next_u32"][5] ^= 1
    doc["resize"]["gray_84x84"][1234] = (doc["resize"]["gray_84x84"][1234] + 1) % 256
    json.dump(doc, open(os.path.join(bad, "upstream_kat.json"), "w"))
    with open(os.path.join(bad, "varstore.pt.tch"), "r+b") as f:   # flip one payload byte of the archive
        blob = f.read()
        f.seek(len(blob) // 2)
        f.write(bytes([blob[len(blob) // 2] ^ 0xFF]))
    report, _ = I.check(bad)
    failed = {item for item, ok, _ in report if not ok}
    assert "rng seed 42" in failed and "border luma (env.rs:176-186)" in failed
    assert any("element 5" in str(d) for item, ok, d in report if item == "rng seed 42")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "ingest_upstream.py"), bad, "--accept"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "MISMATCH rng seed 42" in r.stdout
    assert not os.path.exists(os.path.join(bad, "upstream_pins.json"))


def test_accept_pins_and_flips_the_design_statements_but_never_for_candidate_output(candidate, tmp_path):
    design = str(tmp_path / "DESIGN.md")
    shutil.copyfile(os.path.join(ROOT, "DESIGN.md"), design)
    _, doc = I.check(candidate)
    with pytest.raises(SystemExit):
        I.accept(candidate, doc, design_path=design, golden_dir=str(tmp_path))          # candidate output is not a pin
    real = dict(doc)
    real.pop("producer")
    before = open(design).read()
    have_pins = os.path.exists(I.PINS)
    n = I.accept(candidate, real, design_path=design, golden_dir=str(tmp_path))
    pins = json.load(open(tmp_path / "upstream_pins.json"))
    assert len(pins["rng"]) == 6 and len(pins["varstore"]) == 2 and os.path.exists(tmp_path / "upstream_varstore.pt.tch")
    if not have_pins:      # the repository still says "unconfirmed": all three statements exist and flip
        assert n == 3
        after = open(design).read()
        assert "upstream-unconfirmed" in before and "upstream-unconfirmed" not in after and after.count("pinned") > before.count("pinned")


def test_oracle_reproduces_the_upstream_pins_when_they_exist():
    if not os.path.exists(I.PINS):
        pytest.skip("tools/upstream_kat has not been run on a machine with cargo yet: DESIGN.md section 3 says so")
    from oracle import oracle as O
    pins = json.load(open(I.PINS))
    for rec in pins["rng"]:
        mine = I.oracle_rng_record(O, int(rec["seed"]))
        for k in ("next_u32", "then_next_u64", "then_fill_bytes_13", "then_next_u32", "indices"):
            assert mine[k] == rec[k], (rec["seed"], k)


def test_kat_frame_is_deterministic_and_game_like():
    f = I.kat_frame()
    assert f.shape == (210, 160, 3) and f.dtype == np.uint8
    rows = [y for y in range(210) if y not in (50, 151)]          # (the two noisy rows override everything)
    assert (f[100] == 255).all() and (f[rows, 77] == 255).all()
    assert len(np.unique(f[50])) > 50 and len(np.unique(f[10])) < 20
    import hashlib
    assert hashlib.sha256(f.tobytes()).hexdigest() == hashlib.sha256(I.kat_frame().tobytes()).hexdigest()
