#!/usr/bin/env python3
"""Per-step timeline of a rocprofv3 --kernel-trace database taken with the stream overlap ON: for the steady-state opt
steps, every kernel's start/end relative to the step's first kernel (median over steps), its queue, and the idle gap on
its queue before it - shows what the critical path waits for.

    python tools/rocprof_timeline.py gpurun_out/prof/x_results.db [--first-kernel k_gather] [--skip 30]
"""
import sqlite3
import statistics
import sys

from rocprof_summary import short


def main():
    path = sys.argv[1]
    first = sys.argv[sys.argv.index("--first-kernel") + 1] if "--first-kernel" in sys.argv else "k_gather"
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 30
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select {namecol}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
    rows = [(short(n), s, e, q) for n, s, e, q in db.execute(sel)]
    # split into steps at each launch of the first kernel
    steps, cur = [], None
    for r in rows:
        if r[0].startswith(first):
            if cur:
                steps.append(cur)
            cur = []
        if cur is not None:
            cur.append(r)
    steps = [s for s in steps[skip:-1]]
    n = statistics.mode(len(s) for s in steps)
    seq = statistics.mode(tuple(k[0] for k in s) for s in steps if len(s) == n)   # most common start order
    steps = [s for s in steps if tuple(k[0] for k in s) == seq]
    print(f"# {len(steps)} steady-state steps of {n} kernels; times in us relative to the step's first kernel start (median)\n")
    print("| # | kernel | queue | start | end | dur | gap on queue |")
    print("|---:|---|---:|---:|---:|---:|---:|")
    step_len = statistics.median((s2[0][1] - s1[0][1]) / 1000.0 for s1, s2 in zip(steps, steps[1:]))
    qs = sorted({k[3] for k in steps[0]})
    busy = {q: 0.0 for q in qs}
    for i in range(n):
        st = statistics.median((s[i][1] - s[0][1]) / 1000.0 for s in steps)
        en = statistics.median((s[i][2] - s[0][1]) / 1000.0 for s in steps)
        q = steps[0][i][3]
        prev = [j for j in range(i) if steps[0][j][3] == q]
        gap = statistics.median((s[i][1] - s[prev[-1]][2]) / 1000.0 for s in steps) if prev else float("nan")
        busy[q] += en - st
        print(f"| {i} | {steps[0][i][0][:34]} | {qs.index(q)} | {st:.1f} | {en:.1f} | {en - st:.1f} | {gap:.1f} |")
    print(f"\nstep period (first kernel to next step's first kernel): {step_len:.1f} us; busy per queue: "
          + ", ".join(f"q{qs.index(q)} {b:.1f}" for q, b in busy.items()))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    main()
