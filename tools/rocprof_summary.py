#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace [--pmc ...]) into a per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/xyz.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void bdr::(k_igemm(?:_red|_b3)?)<(.*)>\(", name)
    if m:
        inner = m.group(2)
        pol = re.match(r"(\w+)", inner).group(1)
        g = re.search(r"Geom<([\d, ]+)>", inner)
        extra = ""
        if g:
            d = [int(x) for x in g.group(1).split(",")]
            extra = {(84, 84, 4): "C1", (20, 20, 32): "C2", (9, 9, 64): "C3", (1, 1, 3136): "L1"}.get(tuple(d[:3]), "")
        if m.group(1) == "k_igemm_b3" and not g:   # dense policies (IQN): keep the full name
            return re.sub(r"\(.*", "", name).replace("void ", "")
        return f"{m.group(1)}<{pol}{extra}>"
    return re.sub(r"\(.*", "", name).replace("void ", "")


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    stats = {}
    seen = {}
    for name, s, e in rows:
        k = short(name)
        seen[k] = seen.get(k, 0) + 1
        if seen[k] <= skip:
            continue
        st = stats.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1000.0
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    print(f"# rocprofv3 --kernel-trace summary of `{path}` (durations in us; first {skip} calls of each kernel skipped)\n")
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {v[0]} | {v[1]:.1f} | {v[1] / v[0]:.2f} | {v[2]:.2f} | {v[3]:.2f} | {100 * v[1] / total:.1f} |")
    if "--json" in sys.argv:   # machine-readable twin: bench labels -> average us per launch, stamped with the kernel sources' hash
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out = {}
        for k, v in stats.items():   # kernel name -> bench.py's profile label (bench.KERNEL_LABELS); several kernels of one label add up
            lab = bench.kernel_label(k) or k
            out[lab] = round(out.get(lab, 0.0) + v[1] / v[0], 2)
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump({"kernel_source_sha16": bench.kernel_source_hash(), "source_db": path, "skip_first": skip, "kernels_us": out}, f, indent=1)
    # PMC counters if present
    try:
        pm = db.execute("select name from sqlite_master where name='pmc_events'").fetchall()
        if pm:
            pc = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
            q = db.execute("select * from pmc_events limit 1").fetchall()
            if q:
                print("\n## PMC counters (sum over dispatches / dispatch count)\n")
                kn = "name" if "name" in pc else "kernel_name"
                cn = "counter_name" if "counter_name" in pc else "pmc_name"
                vn = "counter_value" if "counter_value" in pc else "value"
                res = db.execute(f"select {kn}, {cn}, sum({vn}), count(*) from pmc_events group by {kn}, {cn}").fetchall()
                print("| kernel | counter | mean per dispatch | dispatches |")
                print("|---|---|---:|---:|")
                for name, c, s, n in sorted(res, key=lambda r: (short(r[0]), r[1])):
                    print(f"| {short(name)} | {c} | {s / n:.1f} | {n} |")
    except Exception as ex:  # schema differences between rocprofv3 builds
        print(f"\n(pmc table not summarised: {ex})")


if __name__ == "__main__":
    main()
