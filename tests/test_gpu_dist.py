"""The multi-rank launch path of bench.py on hardware: `python bench.py --gpus 2` must spawn its own ranks (the driver's
N = 1 command form) and exchange gradients over RCCL (backend "nccl").  The RCCL test needs two GPUs and skips on a 1-GPU box,
where the same launcher is exercised with both ranks sharing the GPU over gloo (functional check, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"] + list(flags), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_two_ranks_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _run({})
    ex = d["config"]["exchange"]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 256 and d["scaling"] == "weak"
    assert ex["backend"] == "nccl" and ex["ranks_seen"] == 2 and ex["allreduce_bytes_per_step"] > 2.7e8
    assert d["roofline"] is not None and d["cpu_baseline"]["value"] is None


def test_bench_launcher_two_ranks_sharing_one_gpu_over_gloo():
    d = _run({"VITRES_DIST_BACKEND": "gloo"}, "--workload", "ref_tiny", "--batch", "8", "--profile-steps", "0")
    ex = d["config"]["exchange"]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16
    assert ex["backend"] == "gloo" and ex["ranks_seen"] == 2 and ex["ranges"] >= 2 and ex["exposed_ms_per_step"] is not None


def test_bench_launcher_two_ranks_bf16_wire_over_gloo():
    """`bench.py --gpus 2 --wire bf16`: the gradient ranges cross the wire as bf16 (half the bytes of the fp32 exchange)."""
    f32 = _run({"VITRES_DIST_BACKEND": "gloo"}, "--workload", "ref_tiny", "--batch", "8", "--profile-steps", "0")
    b16 = _run({"VITRES_DIST_BACKEND": "gloo"}, "--workload", "ref_tiny", "--batch", "8", "--profile-steps", "0", "--wire", "bf16")
    e32, e16 = f32["config"]["exchange"], b16["config"]["exchange"]
    assert e32["dtype"] == "f32" and e16["dtype"] == "bf16" and 2 * e16["allreduce_bytes_per_step"] == e32["allreduce_bytes_per_step"]
    assert abs(b16["config"]["final_loss"] - f32["config"]["final_loss"]) < 5e-2 * abs(f32["config"]["final_loss"])


def _exchange(backend):
    """Two ranks of tests/exchange_worker.py under torch.distributed.run (127.0.0.1 rendezvous)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VITRES_DIST_BACKEND=backend)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    import socket
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "exchange_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("EXCHANGE ")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0][len("EXCHANGE "):])


def _check_exchange(d, backend):
    assert d["backend"] == backend and d["world"] == 2 and d["finite"]
    assert d["bit_identical"], d            # async per-range all-reduces behind three graphs == one blocking all-reduce
    assert d["ranks_agree"] and d["steps_differ"], d
    assert 0.0 < d["bf16_wire_rel_l2"] < 2.0 ** -8, d       # bf16 on the wire: within bf16 rounding of the fp32 exchange


def test_async_range_exchange_equals_blocking_exchange_over_rccl():
    """The gloo twin below runs everywhere; THIS is the run the stream ordering of RCCL's collectives against the next backward
    graph needs (two GPUs: skipped on the 1-GPU development box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _check_exchange(_exchange("nccl"), "nccl")


def test_async_range_exchange_equals_blocking_exchange_two_ranks_sharing_one_gpu_over_gloo():
    _check_exchange(_exchange("gloo"), "gloo")
