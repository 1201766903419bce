"""Builds libborder_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the .so is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libborder_amd.so")
SOURCES = ["replay.hip", "per.hip", "agent_api.hip", "dqn.hip", "mlp_agents.hip", "sac.hip", "iqn.hip", "comm.hip", "atari_prep.hip", "trainer.hip", "async_trainer.hip"]  # missing files are skipped
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))) + ["../../include/border_amd.h"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS
               if os.path.exists(os.path.join(CSRC, f)))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not (force or _stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(path), *[os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS
                                          if os.path.exists(os.path.join(CSRC, h))]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                   "-Wall", "-Wno-unused-function", "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append(cmd)
        objs.append(obj)
    if jobs:   # the translation units are independent: one hipcc per core (BORDER_AMD_BUILD_JOBS=1 for a serial build)
        from concurrent.futures import ThreadPoolExecutor
        n = max(1, min(len(jobs), int(os.environ.get("BORDER_AMD_BUILD_JOBS", os.cpu_count() or 1))))
        with ThreadPoolExecutor(n) as ex:
            for f in [ex.submit(subprocess.check_call, cmd) for cmd in jobs]:
                f.result()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


HOSTCOMM_LIB = os.path.join(HERE, "libborder_amd_hostcomm.so")


def build_hostcomm_library(verbose: bool = False) -> str:
    """libborder_amd_hostcomm.so: the SAME objects with csrc/comm.hip compiled under -DBDR_COMM_HOST_TRANSPORT - the six librccl entry
    points replaced by a host shared-memory transport (csrc/comm_host_transport.hpp) so that the N > 1 call sequences of comm.hip can run
    with several ranks on ONE GPU (tests/test_gpu_multi.py on a 1-GPU box, bench.py's BDR_BENCH_SHARE_GPU flow test).  A test vehicle:
    loaded only when BORDER_AMD_LIB names it."""
    build_library()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "comm.hip")
    obj = os.path.join(CSRC, "comm_hostcomm.o")
    deps = [src, os.path.join(CSRC, "comm_host_transport.hpp"), os.path.join(CSRC, "common.hpp"), os.path.join(HERE, "..", "include", "border_amd.h")]
    if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
               "-DBDR_COMM_HOST_TRANSPORT", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if not os.path.exists(HOSTCOMM_LIB) or os.path.getmtime(HOSTCOMM_LIB) < max(os.path.getmtime(obj), os.path.getmtime(LIB)):
        objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES if s != "comm.hip" and os.path.exists(os.path.join(CSRC, s))]
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HOSTCOMM_LIB] + objs + [obj, "-ldl", "-lrt"])
    return HOSTCOMM_LIB


EXAMPLES = ("train_dqn_synthetic", "online_loop_atari")


def build_examples() -> str:
    """examples/*: programs in compiled code only, on the C ABI (plain g++, no HIP headers): train_dqn_synthetic (a training run
    with host observations) and online_loop_atari (the one-environment loop with the observation resident in HBM).  Returns the
    first program's path."""
    root = os.path.dirname(HERE)
    exes = []
    for name in EXAMPLES:
        src = os.path.join(root, "examples", name + ".cpp")
        exe = os.path.join(root, "examples", name)
        hdr = os.path.join(root, "include", "border_amd.h")   # (a config struct that grows changes the examples' stack frames)
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(LIB), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-I" + os.path.join(root, "include"), "-L" + HERE, "-lborder_amd",
                                   "-Wl,-rpath,$ORIGIN/../border_amd", "-o", exe])
        exes.append(exe)
    return exes[0]


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_examples())
    print(build_hostcomm_library(verbose=True))
