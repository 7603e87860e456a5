"""world_size-2 gloo tests (CPU) of the batch-sharded sampling path: slicing, the single all-gather and its
ragged variant.  The per-rank 'sampler' is the CPU oracle -- only the sharding/collective logic is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mdt_policy_amd import sharding


def test_shard_bounds_cover_and_balance():
    for total in (1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import mdt_oracle as O
        from tests.helpers import cfg_of, inputs_of, load_fixture, params_of
        meta, _ = load_fixture("g1_tiny_mdtv.npz")
        cfg, P = cfg_of(meta), params_of(meta)
        state, goal, noise = inputs_of(dict(meta, B=total, input_seed=5))
        sig = O.get_sigmas_exponential(2, 0.01, 80.0)
        fn = lambda s, x, g, sg: O.sample_ddim(P, cfg, s, x, g, sg, hoist=True)
        full = fn(state, noise * 80.0, goal, sig)
        got = sharding.sample_sharded(fn, state, noise * 80.0, goal, sig)
        lo, hi = sharding.shard_bounds(total, rank, world)
        local = sharding.sample_sharded(fn, state, noise * 80.0, goal, sig, gather=False)
        ok = (got.shape == full.shape and torch.allclose(got, full, rtol=1e-5, atol=1e-6)
              and torch.allclose(local, full[lo:hi], rtol=1e-5, atol=1e-6))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_sample_sharded_world2_gloo(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


def test_single_process_passthrough():
    fn = lambda s, x, g, sg: x * 2
    x = torch.arange(6.0).reshape(3, 2, 1)
    assert torch.equal(sharding.sample_sharded(fn, {"modality": "lang"}, x, x, None), x * 2)
