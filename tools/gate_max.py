#!/usr/bin/env python3
"""Where the long k_gate launches of a default-schedule kernel trace sit (rocprofv3 --kernel-trace database): for every gate-like kernel
the five longest launches with their ordinal, their start relative to the first kernel of the process, and what ran on the device while they
waited.  usage: gate_max.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
print(f"# long gate launches of `{sys.argv[1]}` ({len(rows)} kernel launches, {(rows[-1][2] - t0) / 1e6:.1f} ms from the first to the last)\n")
for pat in ("k_gate", "k_flag_wait"):
    g = [(i, n, s, e) for i, (n, s, e) in enumerate(rows) if pat in n]
    if not g:
        continue
    ordinal = {i: k for k, (i, _, _, _) in enumerate(g)}
    top = sorted(g, key=lambda r: r[2] - r[3])[:5]
    print(f"## {pat}: {len(g)} launches, median {sorted((e - s) / 1e3 for _, _, s, e in g)[len(g) // 2]:.2f} us\n")
    print("| ordinal | duration us | start (ms after the first kernel) | kernels that ran inside its window |")
    print("|---:|---:|---:|---|")
    for i, n, s, e in top:
        inside = {}
        for n2, s2, e2 in rows:
            if s2 < e and e2 > s and pat not in n2 and "k_signal" not in n2:
                k = n2.split("(")[0].replace("void ", "")[:48]
                inside[k] = inside.get(k, 0) + 1
        what = ", ".join(f"{k} x{v}" for k, v in sorted(inside.items(), key=lambda kv: -kv[1])[:6]) or "(nothing)"
        print(f"| {ordinal[i]} | {(e - s) / 1e3:.1f} | {(s - t0) / 1e6:.2f} | {what} |")
    print()
