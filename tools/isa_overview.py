#!/usr/bin/env python3
"""The memory skeleton of every kernel of a translation unit, read off the gfx950 ISA: global loads (L), stores (S), atomics (AT), MFMAs (M),
barriers (|) and the `s_waitcnt vmcnt(n)` the compiler placed between them (Wv(n)).  This is how round 4 found

  * k loops whose header waited for the NEWEST loads (`vmcnt(12) ... vmcnt(0)` where `vmcnt(28) ... vmcnt(16)` keeps the pipeline full): a
    conditional second step gave the header a predecessor on which the consumed register set is the most recently loaded one;
  * weight-gradient kernels consuming their loads six MFMAs after issuing them (one register set);
  * `S Wv(0) S Wv(0) ...` chains: on gfx9 stores share vmcnt with loads and may be acknowledged out of order, so a prefetched value first
    used behind a store costs a wait on that store (SAC's fused kernels);
  * `L Wv(0) L Wv(0) ...` chains: loads behind a data-dependent test, one round trip each.

    python tools/isa_overview.py border_amd/csrc/dqn.hip [filter] [-D...]      # compiles the device side to assembly (no GPU needed)

`ser=` counts the serialised patterns in a kernel's skeleton; the text is clipped to head ... tail for long kernels.
"""
import os
import re
import subprocess
import sys
import tempfile


def skeleton(body):
    seq = []
    for l in body:
        t = l.strip()
        if t.startswith("s_waitcnt"):
            a = t.replace("s_waitcnt ", "")
            if "vmcnt" in a:
                seq.append("W" + re.sub(r"lgkmcnt\(\d+\)", "", a).replace("vmcnt", "v").replace(" ", ""))
        elif t.startswith("global_load") or t.startswith("buffer_load"):
            seq.append("L")
        elif t.startswith("global_store"):
            seq.append("S")
        elif t.startswith("global_atomic"):
            seq.append("AT")
        elif t.startswith("v_mfma"):
            seq.append("M")
        elif t.startswith("s_barrier"):
            seq.append("|")
    out, prev, cnt = [], None, 0
    for x in seq + [None]:
        if x == prev:
            cnt += 1
        else:
            if prev is not None:
                out.append(prev + (str(cnt) if cnt > 1 else ""))
            prev, cnt = x, 1
    return " ".join(out)


def main():
    src = sys.argv[1]
    flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-I" + os.path.join(root, "include"),
                               "-I" + os.path.join(root, "border_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", asm] + defs, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    names = [l.split(":")[0] for l in lines if l.startswith("_Z") and l.split(";")[0].strip().endswith(":")]
    for name in names:
        if flt and flt not in name:
            continue
        st = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        end = next((i for i in range(st, len(lines)) if "s_endpgm" in lines[i]), None)
        if end is None:
            continue
        txt = skeleton(lines[st:end])
        ser = len(re.findall(r"L Wv\(0\) S", txt)) + len(re.findall(r"S Wv\(0\) S", txt)) + len(re.findall(r"L Wv\(0\) L Wv\(0\)", txt))
        short = re.sub(r"^_ZN\d*(_GLOBAL__N_1\d+|bdr\d*L?\d*)", "", name)[:60]
        print(f"{short:60s} ser={ser:2d}  " + ((txt[:170] + " ... " + txt[-120:]) if len(txt) > 300 else txt))


if __name__ == "__main__":
    main()
