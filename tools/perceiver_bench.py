#!/usr/bin/env python3
"""Time the HIP Perceiver resampler (shipped configuration, 2 x 196 Voltron tokens) at a few batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler

m = PerceiverResampler(dim=384, depth=6, dim_head=64, heads=8, num_latents=3, num_time_embeds=1).cuda().eval()
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 128]:
    x = torch.randn(B, 1, 392, 384, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            m(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = m.flops(1, 392) * B
    print(f"B={B:4d}  {ms:8.3f} ms/forward  {fl / ms / 1e9:7.2f} TFLOP/s  ({fl / B / 1e9:.3f} GFLOP/sample)")
