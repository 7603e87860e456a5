#!/usr/bin/env python3
"""Time forward + backward of the HIP Perceiver resampler (shipped configuration, 2 x 196 tokens) at a training batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = PerceiverResampler(dim=384, depth=6, dim_head=64, heads=8, num_latents=3, num_time_embeds=1).cuda()
x = torch.randn(B, 1, 392, 384, device="cuda")
cot = torch.randn(B, 3, 384, device="cuda")


def step():
    m.zero_grad(set_to_none=True)
    (m(x) * cot).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 3 * m.flops(1, 392) * B
print(f"perceiver fwd+bwd B={B}: {dt * 1e3:.2f} ms  ~{fl / dt / 1e12:.1f} TFLOP/s (3x forward FLOPs)  grad ok: "
      f"{all(torch.isfinite(p.grad).all().item() for p in m.parameters())}")
