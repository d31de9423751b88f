"""bench.py as the driver launches it at N > 1 -- `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, one
process per GPU over RCCL -- on min(device_count, 2) real devices, and the single-process form (`--comm`).  A gpurun box has one GPU:
skipped there (the same control flow runs over gloo with the emulated kernels in tests/test_bench_emulated.py); on a node with more,
this is the code's first contact with RCCL at N > 1 before the SCALE run."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_torchrun_two_gpus_matches_one_gpu():
    if _devices() < 2:
        pytest.skip("one GPU visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--proofs", "20000", "--no-cpu-baseline", "--max-seconds", "600"]
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", *common], cwd=ROOT, env=env,
                        capture_output=True, text=True, timeout=900)
    assert p2.returncode == 0, p2.stderr[-3000:]
    two = _line(p2.stdout)
    assert "error" not in two and two["n_gpus"] == 2
    assert two["rccl_world"]["ranks"] == 2 and two["rccl_world"]["distinct_devices"] == 2 and two["rccl_world"]["backend"] == "nccl"
    assert two["roofline"]["verdict_exchange"]["allreduces_on_this_rank"] > 0
    # (bench.py itself asserts that every verdict row equals the failures over BOTH ranks' shards: rc 0 says they did)
    p1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1", *common], cwd=ROOT, env=env, capture_output=True, text=True,
                        timeout=900)
    assert p1.returncode == 0, p1.stderr[-3000:]
    one = _line(p1.stdout)
    # weak leg: the same per-GPU work; strong leg: the same block witness, whoever holds which shard
    assert two["config"]["units_per_gpu_per_step"] == one["config"]["units_per_gpu_per_step"]
    assert two["strong"]["nodes_shipped"] > 0 and two["strong"]["predicted"]["n_gpus"] == 2
    assert two["strong"]["workload"].split(" split over")[0] == one["strong"]["workload"].split(" split over")[0]


@pytest.mark.gpu
def test_two_ranks_one_device_gloo():
    """What a ONE-GPU box can say about the process-per-GPU form: two ranks under torch.distributed.run, both on device 0, the
    verdict all-reduce over gloo (staged through host memory) -- the shard slices, the per-pass verdict exchange on device
    tensors, the `strong` leg and bench.py's own assertion that every verdict row equals the failures over BOTH ranks' shards,
    over the HIP kernels instead of the emulator.  Not a scaling number, and the line says so."""
    if _devices() < 1:
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PHANT_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--one-device", "--steps", "2",
                        "--warmup", "1", "--proofs", "20000", "--no-cpu-baseline", "--max-seconds", "600"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    two = _line(p.stdout)
    assert "error" not in two and two["n_gpus"] == 2 and two["value"] > 0
    rw = two["rccl_world"]
    assert rw["ranks"] == 2 and rw["backend"] == "gloo" and rw["distinct_devices"] == 1
    assert two["roofline"]["verdict_exchange"]["allreduces_on_this_rank"] > 0
    assert two["strong"]["predicted"]["n_gpus"] == 2 and two["strong"]["value"] > 0
    # the same control flow for node-set witnesses
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--one-device", "--workload",
                        "nodeset", "--steps", "2", "--warmup", "1", "--proofs", "20000", "--no-cpu-baseline", "--max-seconds", "600"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    ns = _line(p.stdout)
    assert ns["n_gpus"] == 2 and ns["rccl_world"]["distinct_devices"] == 1 and ns["value"] > 0


@pytest.mark.gpu
def test_comm_form_two_gpus():
    if _devices() < 2:
        pytest.skip("one GPU visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "bench.py", "--comm", "--comm-devices", "2", "--steps", "2", "--warmup", "1", "--max-seconds",
                        "600"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _line(p.stdout)
    assert line["n_gpus"] == 2 and len(set(line["devices"])) == 2 and line["value"] > 0


@pytest.mark.gpu
def test_the_wall_clock_guard_prints_an_error_line():
    """One GPU is enough for this one: a run that cannot finish in --max-seconds ends with a parseable line and rc 4."""
    if _devices() < 1:
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2000", "--inner", "50", "--max-seconds", "20", "--no-cpu-baseline",
                        "--no-strong"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 4, (p.returncode, p.stderr[-1500:])
    line = _line(p.stdout)
    assert line["value"] is None and "wall-clock guard" in line["error"]
