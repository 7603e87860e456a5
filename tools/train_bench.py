#!/usr/bin/env python3
"""Time the HIP training step (GCDenoiser.loss forward + backward, then torch AdamW) of the MDT-V default model
at the reference's per-GPU batch size (conf/config.yaml:23 batch_size 128), eval-mode and train-mode (dropout)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import configs, synthetic
from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = configs.mdtv_default()
model = GCDenoiser(cfg, 0.5).cuda()
inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
state = {"state_images": inp["state_images"], "modality": "lang"}
if os.environ.get("MDT_TRAIN_BENCH_OPT", "torch") == "fused":  # this package's one-launch AdamW (what bench.py's step uses)
    from mdt_policy_amd.optim import FusedAdamW
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
else:
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
n_params = sum(p.numel() for p in model.parameters())
for mode in os.environ.get("MDT_TRAIN_BENCH_MODES", "eval,train").split(","):
    model.train(mode == "train")

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    eng = model.inner_model.hip_engine()
    fwd_flops = eng.flops_per_chunk(1) * B          # one model evaluation per sample
    print(f"{mode:5s} B={B}: {dt * 1e3:8.2f} ms/step  {B / dt:9.1f} samples/s  ~{3 * fwd_flops / dt / 1e12:6.2f} TFLOP/s "
          f"(3x forward FLOPs)  params {n_params / 1e6:.1f} M  loss {loss.item():.4f}")
