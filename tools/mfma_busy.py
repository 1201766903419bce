#!/usr/bin/env python3
"""MFMA-busy fraction per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_MFMA; --kernel-trace only).

    python tools/mfma_busy.py <bench pass .db> <calibration pass .db> > profiles/rocprof_rNN_mfma_busy.md

rocprofv3 reports the SQ counter in several rows per dispatch and in its own units; instead of guessing the normaliser, the same
counter pair is collected for tools/probes/mfma_busy_cal.bin, whose kernels issue MFMAs back to back on every SIMD: the ratio
sum(SQ_VALU_MFMA_BUSY_CYCLES) / sum(GRBM_GUI_ACTIVE) of those launches is what "100 % busy" reads as (one value per MFMA shape:
FP32 32x32x2 for the k_igemm / dense kernels, bf16 32x32x16 for conv1 and the split-operand kernels).  busy = ratio / ratio_cal.
"""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocprof_summary import short  # noqa: E402


def totals(path):
    db = sqlite3.connect(path)
    pc = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
    kn = "name" if "name" in pc else "kernel_name"
    cn = "counter_name" if "counter_name" in pc else "pmc_name"
    vn = "counter_value" if "counter_value" in pc else "value"
    out = {}
    for name, c, s, n in db.execute(f"select {kn}, {cn}, sum({vn}), count(*) from pmc_events group by {kn}, {cn}"):
        d = out.setdefault(short(name), {})
        d[c] = d.get(c, 0.0) + s
        d[c + "#rows"] = d.get(c + "#rows", 0) + n
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    for name, n in db.execute(f"select {namecol}, count(*) from kernels group by {namecol}"):
        if short(name) in out:
            out[short(name)]["#dispatches"] = out[short(name)].get("#dispatches", 0) + n
    return out


def main():
    run, cal = totals(sys.argv[1]), totals(sys.argv[2])
    ratio = lambda d: d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["GRBM_GUI_ACTIVE"]
    c32 = ratio(next(v for k, v in cal.items() if "mfma_cal_f32" in k))
    c16 = ratio(next(v for k, v in cal.items() if "mfma_cal_bf16" in k))
    print("# MFMA-busy per kernel: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, normalised by the same ratio of a back-to-back MFMA kernel\n")
    print(f"calibration (`tools/probes/mfma_busy_cal.bin` under the same rocprofv3 pass): FP32 32x32x2 ratio {c32:.3f}, bf16 32x32x16 ratio {c16:.3f} = 100 % busy\n")
    print("| kernel | dispatches | MFMA instructions / dispatch | busy cycles / GUI-active cycle | MFMA busy (of the launch's GUI-active time) | pipe |")
    print("|---|---:|---:|---:|---:|---|")
    rows = []
    for k, d in run.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d or d["SQ_VALU_MFMA_BUSY_CYCLES"] == 0:
            continue
        bf = "conv1" in k or "b3" in k
        n = d.get("#dispatches") or d["GRBM_GUI_ACTIVE#rows"]
        rows.append((ratio(d) / (c16 if bf else c32), k, n, d.get("SQ_INSTS_MFMA", 0.0) / n, ratio(d), "bf16" if bf else "fp32"))
    for busy, k, n, inst, r, pipe in sorted(rows, reverse=True):
        print(f"| {k} | {n} | {inst:.0f} | {r:.3f} | {busy:.3f} | {pipe} |")


if __name__ == "__main__":
    main()
