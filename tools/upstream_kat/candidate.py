"""Writes what tools/upstream_kat/src/main.rs writes - same files, same JSON layout - but from border_amd's OWN oracle
(oracle.StdRng, oracle/atari_prep.py) and the library's own checkpoint writer: the *candidate* values the real crates are
expected to reproduce.  Used by tests/test_upstream_kat.py to exercise tests/golden/ingest_upstream.py end to end before any
machine with cargo has run the Rust program; its output is never accepted as a pin (the JSON says "producer": "candidate")."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import ingest_upstream as I  # noqa: E402


def main(out):
    from oracle import atari_prep as P
    from oracle import oracle as O
    from border_amd import checkpoint as CK
    os.makedirs(out, exist_ok=True)
    doc = {"format": 1, "producer": "candidate", "crates": {"rand": "0.8.5", "tch": "0.16", "image": "0.23.14"}, "rng": []}
    for seed in I.SEEDS:
        doc["rng"].append(I.oracle_rng_record(O, seed))
    tensors = I.varstore_tensors()
    for f in I.VARSTORE_FILES:
        CK.write(os.path.join(out, f), tensors)
    doc["varstore"] = {"files": list(I.VARSTORE_FILES), "tensors": [[n, list(s), o] for n, s, o in I.VARSTORE_SPEC],
                       "formula": "value[i] = i * 0.25 - 3 + offset"}
    frame = I.kat_frame()
    rgb = P.resize_triangle(frame, 84, 84)
    doc["resize"] = {"width": 160, "height": 210, "rgb_84x84": rgb.ravel().tolist(), "gray_84x84": P.grayscale(rgb).ravel().tolist()}
    with open(os.path.join(out, "upstream_kat.json"), "w") as f:
        json.dump(doc, f)
    print("wrote", os.path.join(out, "upstream_kat.json"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "out_candidate")
