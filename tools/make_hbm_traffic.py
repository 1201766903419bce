#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 --pmc summaries (FETCH_SIZE pass, WRITE_SIZE pass).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; on gfx950 FETCH_SIZE counts
half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section) - calibrated here on
k_gather (14.45 MB read, 14.45 MB written by construction) and the full-arena k_adam of earlier rounds (27.0 MB read, 20.2 MB written).
usage: make_hbm_traffic.py fetch.md write.md > profiles/hbm_traffic.json"""
import json
import re
import sys

KEY = {"bdr::k_conv1_bf16": "fwd_conv1", "k_adam": "adam_l1_l2", "k_gather": "sample", "k_igemm<DxC2P>": "bwd_conv2_dx",
       "k_igemm<DxC3P": "bwd_conv3_dx", "k_igemm<DxL1>": "bwd_l1_dx", "k_igemm<FwdL1>": "fwd_l1", "k_igemm<FwdL1Z2>": "fwd_l1", "k_igemm<FwdPC2>": "fwd_conv2",
       "k_igemm<FwdPC3>": "fwd_conv3", "bdr::k_conv1_dw_bf16": "bwd_conv1_dw", "k_igemm_red<DwPC1>": "bwd_conv1_dw", "k_igemm_red<DwPC2>": "bwd_conv2_dw",
       "k_igemm_red<DwPC3>": "bwd_conv3_dw", "k_igemm_red<DwPL1>": "bwd_l1_dw", "k_reduce_partials3": "bwd_conv_reduce", "k_reduce_adam": "reduce_adam",
       "k_head<": "head_fwd_td", "k_head_fwd": "head_fwd", "k_head_bwd": "head_bwd", "k_td_rows": "td_rows"}


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\| (.+?) \| %s \| ([0-9.]+) \|" % counter, line)
        if m:
            name = m.group(1).strip()
            for k, v in KEY.items():
                if name.startswith(k):
                    out[v] = float(m.group(2))
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
