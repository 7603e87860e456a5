"""Determinism soak of the rollout-sized paths (round 5: the MFMA attention of k_attn_proj_smallm works in wave-local LDS regions
without a workgroup barrier, side jobs ride in neighbouring launches): N calls per batch size, eager and graph-replayed, with NEW
noise every few calls; every call must reproduce, bit for bit, what the first call with the same inputs returned."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mdt_policy_amd import synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
dev = torch.device("cuda")
cfg, P, model = bench.build_model(dev)
sig = gs.get_sigmas_exponential(10, 0.001, 80.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = 0
for mode in ("0", "1"):
    gs._GRAPH_MODE = mode; gs._GRAPH_SAMPLER = mode == "1"
    for B in (1, 2, 3, 5, 8, 16):
        ref = {}
        with torch.no_grad():
            for i in range(N):
                seed = i % 7
                inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, seed=seed).items()}
                st = {"state_images": inp["state_images"], "modality": "lang" if seed % 2 else "vis"}
                out = gs.sample_ddim(model, st, inp["noise"] * 80, inp["goal"], sig).clone()
                if seed not in ref: ref[seed] = out
                elif not torch.equal(out, ref[seed]): bad += 1
        torch.cuda.synchronize()
        print(f"graph={mode} B={B:3d}: {N} calls, mismatches so far {bad}", flush=True)
assert bad == 0
print("rollout soak ok")
