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
    assert len(j["collective"]["per_rank_ms"]) == 2 and j["collective"]["gather_us"] > 0


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_bench_two_ranks_sharing_one_gpu_over_gloo():
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, barrier + synchronize around the timed
    region, MAX over ranks, one gather of the sampled actions per call, rank 0 prints ONE JSON line) on a box with a single GPU:
    both ranks use device 0 and gather through gloo (MDT_BENCH_SHARE_GPU / MDT_BENCH_BACKEND; the numbers mean nothing)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MDT_BENCH_VERIFY_GATHER="1", MDT_BENCH_SHARE_GPU="1", MDT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "32", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the bench line"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 2 and j["warmup"] == 1
    assert j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"] == 64
    assert j["collective"]["backend"] == "gloo" and j["collective"]["ranks"] == 2 and j["collective"]["rccl_ranks"] == 0
    assert j["collective"]["gather_verified"] is True
    assert j["value"] > 0 and abs(j["value"] - 64 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]
    # the N > 1 line says where a shortfall comes from: the collective's time on every rank's stream, every rank's own step time
    c = j["collective"]
    assert c["bytes_per_rank"] == 32 * 10 * 7 * 4
    assert len(c["per_rank_ms"]) == len(c["per_rank_gpu_ms"]) == len(c["gather_us_per_rank"]) == len(c["gather_us_max_per_rank"]) == 2
    assert all(v > 0 for v in c["per_rank_ms"] + c["per_rank_gpu_ms"] + c["gather_us_per_rank"])
    assert c["gather_us"] == min(c["gather_us_per_rank"]) and c["slowest_rank"] in (0, 1)
    assert max(c["per_rank_ms"]) <= j["ms_per_step"] * 1.001  # the quoted step time is the slowest rank's, barrier included


_RAGGED_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MDT_ROOT"])
from mdt_policy_amd import configs, sharding, synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
cfg = configs.mdtv_tiny()
model = GCDenoiser(cfg, 0.5)
shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=3, profile="rich").items()})
model = model.to(dev).eval()
total = 2 * 8 + 1                                   # 17 chunks over 2 ranks: shards of 9 and 8 -> the padded (ragged) gather
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(total, cfg, seed=4).items()}   # replicated request
state = {"state_images": inp["state_images"], "modality": "lang"}
sig = gs.get_sigmas_exponential(4, 0.001, 80.0)
fn = lambda s, x, g, sg: gs.sample_ddim(model, s, x, g, sg)
with torch.no_grad():
    got = sharding.sample_sharded(fn, state, inp["noise"] * 80.0, inp["goal"], sig)
    whole = fn(state, inp["noise"] * 80.0, inp["goal"], sig)                  # every rank also samples the whole request
torch.cuda.synchronize()
lo, hi = sharding.shard_bounds(total, rank, world)
ok = got.shape == whole.shape and torch.allclose(got, whole, rtol=1e-4, atol=1e-5) and hi - lo == (9 if rank == 0 else 8)
flag = torch.tensor([1 if ok else 0], device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("RAGGED_OK" if flag.item() == 1 else "RAGGED_BAD", dist.get_backend(), tuple(got.shape), flush=True)
dist.destroy_process_group()
'''


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_ragged_shards_over_rccl(tmp_path):
    """sharding.sample_sharded with a request that does not divide by the ranks (17 chunks, 2 ranks): the padded
    all_gather_into_tensor branch over RCCL -- the other collective shape of the multi-GPU path (the equal-shard one is
    covered by the bench launch above; both run on gloo in tests/test_sharding_gloo.py)."""
    script = tmp_path / "ragged_worker.py"
    script.write_text(_RAGGED_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MDT_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "RAGGED_OK nccl (17, 10, 7)" in res.stdout, res.stdout[-500:]
