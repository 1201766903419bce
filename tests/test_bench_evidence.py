"""bench.py's `roofline` object says where its numbers come from: `traffic` (PMC-derived HBM bytes per launch) is reported only when
profiles/hbm_traffic.json was measured on the kernel sources the running library is built from, otherwise null with a note; the
inputs of `achieved` (kernel_gflop, launches, instances) and the rocprofv3 twin of the HIP-event time sit in the same object."""
import json
import os

import bench


def _tree(tmp_path, traffic_hash_matches, monkeypatch):
    root = tmp_path / "repo"
    (root / "border_amd" / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    (root / "profiles").mkdir()
    (root / "border_amd" / "csrc" / "a.hip").write_text("__global__ void k() {}\n")
    (root / "border_amd" / "csrc" / "b.hpp").write_text("// header\n")
    (root / "include" / "border_amd.h").write_text("/* abi */\n")
    monkeypatch.setattr(bench, "ROOT", str(root))
    h = bench.kernel_source_hash()
    meta = {"kernel_source_sha16": h if traffic_hash_matches else "0" * 16, "commit": "abc1234", "pmc_tables": "profiles/rocprof_r03_pmc.md"}
    (root / "profiles" / "hbm_traffic.json").write_text(json.dumps({"fwd_conv2": 42938982, "_source": meta}))
    (root / "profiles" / "kernel_trace_c2_serial.json").write_text(json.dumps({"kernel_source_sha16": h, "kernels_us": {"fwd_conv2": 32.1}}))
    return root, h


def _roof():
    fl = bench.dqn_kernel_flops(256, 2)
    conf = {"flops": fl, "bytes": {"sample": 14454272}, "step_flops": sum(fl.values()), "batch": 256, "name": "c2",
            "flops_per_instance": bench.dqn_kernel_flops(256, 1)}
    prof = {k: 0.02 for k in fl}
    prof["fwd_conv2"] = 0.0307
    prof["sample"] = 0.009
    cnt = {k: 1 for k in prof}
    return bench.roofline(conf, prof, cnt, 0.0026, 0.221)


def test_traffic_is_reported_with_its_source_when_the_sources_match(tmp_path, monkeypatch):
    _, h = _tree(tmp_path, True, monkeypatch)
    r = _roof()
    assert r["kernel"] == "fwd_conv2" and r["traffic"] == 42938982
    assert r["traffic_source"]["kernel_source_sha16"] == h == r["kernel_source_sha16"] and r["traffic_source"]["pmc_tables"].endswith("pmc.md")
    # everything `achieved` is computed from, in the object itself
    assert abs(r["kernel_gflop"] - 2.7181) < 1e-3 and r["units_per_launch"] == {"launches_per_step": 1, "batch_rows": 256, "network_instances": 2}
    assert abs(r["achieved"] - r["kernel_gflop"] / r["kernel_ms"]) < 0.05            # GFLOP / ms == TFLOP/s
    assert abs(r["frac"] - r["achieved"] / 157.3) < 1e-3
    assert r["rocprof_check"]["rocprofv3_avg_us"] == 32.1 and r["rocprof_check"]["same_kernel_sources"] is True
    assert abs(r["rocprof_check"]["hip_event_us"] - 30.7) < 0.01


def test_stale_traffic_is_withheld(tmp_path, monkeypatch):
    root, h = _tree(tmp_path, False, monkeypatch)
    r = _roof()
    assert r["traffic"] is None and r["traffic_source"]["stale"] is True and h in r["traffic_source"]["note"]
    # ... and any edit of a kernel source changes the hash
    (root / "border_amd" / "csrc" / "a.hip").write_text("__global__ void k() { }\n")
    assert bench.kernel_source_hash() != h


def test_fp32_gemm_sum_and_traffic_ratios_are_in_the_line(tmp_path, monkeypatch):
    """What the round-3 review had to recompute by hand: all FP32-MFMA GEMM launches together on the rocprofv3 durations of the
    committed trace, and measured / algorithmic HBM bytes for every kernel with a byte model."""
    root, h = _tree(tmp_path, True, monkeypatch)
    fl = bench.dqn_kernel_flops(256, 2)
    ab = bench.dqn_kernel_bytes(256, 2)
    us = {k: 20.0 for k in fl}
    (root / "profiles" / "kernel_trace_c2_serial.json").write_text(json.dumps({"kernel_source_sha16": h, "kernels_us": us}))
    (root / "profiles" / "hbm_traffic.json").write_text(json.dumps({"fwd_conv2": 42938982, "bwd_conv2_dx": 2 * ab["bwd_conv2_dx"], "fwd_l1": 45964288,
                                                                     "_source": {"kernel_source_sha16": h, "pmc_tables": "profiles/x_pmc.md"}}))
    conf = {"flops": fl, "bytes": {"sample": 14454272}, "step_flops": sum(fl.values()), "batch": 256, "name": "c2",
            "flops_per_instance": bench.dqn_kernel_flops(256, 1), "alg_bytes": ab}
    prof = {k: 0.02 for k in fl}
    r = bench.roofline(conf, prof, {k: 1 for k in prof}, 0.0026, 0.221)
    g = r["fp32_gemm_sum"]
    fp32 = [k for k in fl if k not in bench.BF16_ISSUE]
    assert g["kernels"] == sorted(fp32) and abs(g["rocprofv3_us"] - 20.0 * len(fp32)) < 1e-6 and g["same_kernel_sources"] is True
    assert abs(g["gflop"] - sum(fl[k] for k in fp32) / 1e9) < 1e-2 and abs(g["frac"] - g["gflop"] / (g["rocprofv3_us"] * 1e-3) / 157.3) < 1e-3
    t = r["traffic_vs_algorithmic"]
    assert t["bwd_conv2_dx"]["ratio"] == 2.0 and t["fwd_l1"]["alg_bytes"] == ab["fwd_l1"] and "fwd_conv3" not in t
    # conv2 forward, both networks: 2 x (a1 13.1 MB + W2 0.13 MB + a2 5.3 MB)
    assert ab["fwd_conv2"] == 2 * (256 * 400 * 32 * 4 + 512 * 64 * 4 + 256 * 81 * 64 * 4)
    # a PMC pass of other kernel sources is withheld here too
    (root / "profiles" / "hbm_traffic.json").write_text(json.dumps({"fwd_conv2": 1, "_source": {"kernel_source_sha16": "0" * 16}}))
    assert bench.roofline(conf, prof, {k: 1 for k in prof}, 0.0026, 0.221)["traffic_vs_algorithmic"] == {"stale": True}


def test_committed_kernel_trace_and_pmc_pass_are_of_the_same_kernel_sources():
    """The evidence pack of a round is ONE set of kernel sources: profiles/kernel_trace_c2_serial.json (rocprofv3 durations) and
    profiles/hbm_traffic.json (PMC bytes) must name the same source hash, or the per-kernel table mixes two binaries."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kt = json.load(open(os.path.join(root, "profiles", "kernel_trace_c2_serial.json")))
    tr = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
    assert kt["kernel_source_sha16"] == tr["_source"]["kernel_source_sha16"], (kt["kernel_source_sha16"], tr["_source"])


def test_committed_evidence_files_are_stamped():
    """What is committed under profiles/ names the sources it was measured on (it may be stale - then bench says so - but never unlabelled)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
    assert "_source" in doc and len(doc["_source"]["kernel_source_sha16"]) == 16


def test_the_dominant_kernel_of_the_line_does_not_flip_between_near_ties():
    """C2's two longest launches (conv2 forward of both networks, conv2's input gradient) are within a few % of each other and trade
    places from run to run; the line names the one with the most algorithmic work among launches within 5 % of the longest, and lists
    the other beside it - a different kernel only when it really is longer."""
    import bench
    conf = {"name": "c2", "flops": {"fwd_conv2": 2.718e9, "bwd_conv2_dx": 1.359e9, "fwd_conv3": 1.85e9}, "bytes": {"sample": 14454272},
            "step_flops": 17.46e9, "batch": 256}
    cnt = {"fwd_conv2": 1, "bwd_conv2_dx": 1, "fwd_conv3": 1, "sample": 1}
    r = bench.roofline(conf, {"fwd_conv2": 0.0299, "bwd_conv2_dx": 0.0301, "fwd_conv3": 0.022, "sample": 0.011}, cnt, 0.003, 0.22)
    assert r["kernel"] == "fwd_conv2" and list(r["within_5pct"]) == ["bwd_conv2_dx"]
    assert abs(r["frac"] - 2.718e9 / 0.0299e-3 / 1e12 / bench.PEAK_FP32_MFMA_TFLOPS) < 1e-3
    r = bench.roofline(conf, {"fwd_conv2": 0.0299, "bwd_conv2_dx": 0.0330, "fwd_conv3": 0.022, "sample": 0.011}, cnt, 0.003, 0.22)
    assert r["kernel"] == "bwd_conv2_dx" and "within_5pct" not in r


def test_the_window_protocol_is_frozen_with_its_version():
    """The legs of the default bench run and their order define what `value` means.  Round 4 re-ordered them once (version 1 -> 2, disclosed in
    the line); from here on a change of `window_order` without a `protocol_version` bump fails this test, so a BENCH_rNN.json series cannot be
    re-defined silently.  `value_cold` (version 1's `value`) and `value_steady` are first-class keys of the line."""
    import hashlib
    import inspect
    frozen = {2: "4910ceb1b087"}   # protocol_version -> sha256(window_order)[:12]
    assert bench.PROTOCOL_VERSION in frozen, "a new protocol version: add its window-order hash here, and say so in DESIGN.md section 6"
    assert hashlib.sha256(bench.WINDOW_ORDER.encode()).hexdigest()[:12] == frozen[bench.PROTOCOL_VERSION], \
        "bench.WINDOW_ORDER changed without a PROTOCOL_VERSION bump"
    src = inspect.getsource(bench.main)
    for key in ('"value_cold"', '"value_steady"', '"protocol_version"', '"warmup_note"', '"window_order"', '"untimed_steps_before_window"'):
        assert key in src, key
    # the legs run in the order the string names: cold window, steady-state loop, the timed window
    i_cold, i_ss, i_val = src.index("dt_c = window(0)"), src.index("run(n_ss,"), src.index("dt = window(0 if cold is None")
    assert i_cold < i_ss < i_val


def test_c4_step_work_counts_every_layer_once(tmp_path, monkeypatch):
    """Round 5's C4 lines printed step.gflop 936.7 (the four `*_3xbf16` aliases were summed with their layers) and a step fraction of
    1.23 "of the FP32 peak" for the exact-f32 run.  The step's work is the sum over DISTINCT layers (489.5 GFLOP at batch 512, 64
    quantiles), and a step whose kernels all sit on the FP32 pipe cannot exceed its peak."""
    _tree(tmp_path, True, monkeypatch)
    fl = bench.iqn_kernel_flops(512, 64)
    step_flops = sum(fl.values())
    assert abs(step_flops / 1e9 - 489.5) < 0.1
    aliased = dict(fl)
    aliased.update({k + "_3xbf16": fl[k] for k in bench.C4_SPLIT_LAYERS})
    assert bench.distinct_layer_flops(aliased) == step_flops and sum(aliased.values()) > 1.9 * step_flops   # (what round 5 summed)
    # an exact-f32 run: every label of the work model on the FP32 pipe, each launch AT the pipe's peak, serial schedule
    saved = dict(bench.BF16_ISSUE)
    try:
        bench.BF16_ISSUE.clear()
        prof = {k: v / (bench.PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3 for k, v in fl.items()}
        conf = {"flops": aliased, "bytes": {"sample": 1}, "step_flops": step_flops, "batch": 512, "name": "c4"}
        r = bench.roofline(conf, prof, {k: 1 for k in prof}, 0.0, sum(prof.values()))
    finally:
        bench.BF16_ISSUE.update(saved)
    assert abs(r["step"]["gflop"] - step_flops / 1e9) < 1e-2
    assert r["step"]["frac"] <= 1.0 + 1e-3 and abs(r["step"]["fp32_mfma"]["gflop"] - r["step"]["gflop"]) < 1e-2
