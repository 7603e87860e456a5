"""Masked generative foresight head alone: forward + backward time at training batch sizes (fp32, synthetic images).
usage: python tools/mae_bench.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd.models.img_generation.masked_transformer_decoder import MaskedTransformerImgDecoder


def mae_flops(B, ctx_tokens=4, d=192, depth=6, n=49, keep=12, pdim=768, cdim=384):
    T = ctx_tokens + 2 * n
    blk = T * 2 * (d * 3 * d + d * d + d * 8 * d + 4 * d * d) + 4 * T * T * d
    fwd = ctx_tokens * 2 * cdim * d + 2 * keep * 2 * pdim * d + depth * blk + 2 * n * 2 * d * pdim
    return 3.0 * fwd * B


if __name__ == "__main__":
    dev = torch.device("cuda")
    torch.manual_seed(0)
    gen = MaskedTransformerImgDecoder(112, 16, 6, 192, 8, 384, mask_ratio=0.75).to(dev)
    for B in [int(x) for x in (sys.argv[1:] or ["128", "1024"])]:
        ctx = torch.randn(B, 4, 384, device=dev, requires_grad=True)
        img = torch.randn(B, 2, 3, 112, 112, device=dev)

        def step():
            for p in gen.parameters():
                p.grad = None
            rec, mask, restore, _ = gen(ctx, img)
            gen.compute_loss(img, rec, mask, restore).backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"B={B:5d}: {dt*1e3:8.2f} ms forward+backward, {mae_flops(B)/dt/1e12:6.1f} TFLOP/s "
              f"({mae_flops(B)/dt/1e12/157.3*100:.1f} % of the fp32-MFMA peak)", flush=True)
