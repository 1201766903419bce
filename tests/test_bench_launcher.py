"""`python bench.py --gpus N` as a plain command (no torchrun around it, no WORLD_SIZE in the environment) launches its own N
ranks (bench.self_launch -> torch.distributed.run on 127.0.0.1) and prints ONE JSON line from rank 0; the driver's form
(`python -m torch.distributed.run ... bench.py --gpus N`) runs the ranks it is given.  --dry-run stops before the first GPU call,
so the launcher and the control plane (gloo rendezvous, barrier, MAX-reduce) are covered in the CPU-only container."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GPU_MAX_HW_QUEUES")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_plain_command_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["steps"] == 3 and line["warmup"] == 1
    assert line["gpu_max_hw_queues"] == "8"          # the N>1 path asks for its hardware queues before the first HIP call


def test_eight_ranks_the_size_of_config_c3():
    """BASELINE config 3 is 8 actors x 8 MI355X: the launcher, the gloo rendezvous on 127.0.0.1, the barrier and the MAX-reduce of
    the timing with eight ranks (no GPU is touched: --dry-run)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 8 and line["ranks_seen"] == list(range(8)) and line["gpu_max_hw_queues"] == "8"


def test_driver_form_under_torchrun_runs_the_ranks_it_is_given():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == [0, 1]


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["ranks_seen"] == [0]


def test_a_failing_rank_gives_a_nonzero_status():
    """No GPU in this container: without --dry-run every rank exits non-zero ('HIP device not visible'), and so does the launcher."""
    import border_amd
    if border_amd.device_count() > 0:
        import pytest
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
