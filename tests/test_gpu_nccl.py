"""The N > 1 path with RCCL itself (backend "nccl" on ROCm): runs only where at least two GPUs are visible -- the driver's
multi-GPU node -- and is skipped on the 1-GPU boxes.  bench.py is launched exactly as the driver launches it."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_bench_two_ranks_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MDT_BENCH_VERIFY_GATHER="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"]
    assert j["collective"]["backend"] == "nccl" and j["collective"]["rccl_ranks"] == 2
    assert j["collective"]["gather_verified"] is True  # shape (2 B, 10, 7), rank-ordered blocks equal to each rank's own output
    assert j["value"] > 0
