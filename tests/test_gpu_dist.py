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
