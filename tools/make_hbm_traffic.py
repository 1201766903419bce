#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 --pmc summaries (FETCH_SIZE pass, WRITE_SIZE pass).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; on gfx950 FETCH_SIZE counts
half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section) - calibrated here on
k_gather (14.45 MB read, 14.45 MB written by construction) and the full-arena k_adam of earlier rounds (27.0 MB read, 20.2 MB written).
usage: make_hbm_traffic.py fetch.md write.md > profiles/hbm_traffic.json"""
import json
import re
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402  (kernel name -> profile label: bench.KERNEL_LABELS)


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\| (.+?) \| %s \| ([0-9.]+) \|" % counter, line)
        if m:
            lab = bench.kernel_label(m.group(1).strip())
            if lab:
                out[lab] = float(m.group(2))
    return out


import os
import subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
res = {k: int(round((2 * fetch[k] + write.get(k, 0.0)) * 1024)) for k in fetch}
res["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc (separate passes, B=256, "
                "2 network instances per forward launch, BDR_NO_OVERLAP=1). gfx950 correction per MI355X_MICROARCH.md "
                "(FETCH_SIZE counts half of wide coalesced reads), calibrated on k_gather (algorithmic 14.45 MB read / "
                "14.45 MB written) and the full-arena Adam pass (27.0 MB / 20.2 MB). Source tables: the pmc_tables file named in _source")
# which binary this was measured on: bench.py withholds `roofline.traffic` when the kernel sources have changed since
import bench
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    commit = None
res["_source"] = {"kernel_source_sha16": os.environ.get("KERNEL_SOURCE_SHA16") or bench.kernel_source_hash(), "commit": commit,
                  "pmc_tables": sys.argv[3] if len(sys.argv) > 3 else None, "fetch_pass": sys.argv[1], "write_pass": sys.argv[2]}
print(json.dumps(res, indent=1))
