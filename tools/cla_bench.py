"""Time the contrastive head: ClipStyleProjection('map') forward / forward+backward and the InfoNCE op (value + gradients)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd.models.contrastive import clip_auxiliary_loss
from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


clip = ClipStyleProjection("map", 384, 1, 4).cuda()
for B in (128, 1024):
    x = torch.randn(B, 4, 384, device="cuda")
    with torch.no_grad():
        f = timed(lambda: clip(x))
    xg = x.clone().requires_grad_()
    def fb():
        clip.zero_grad(set_to_none=True)
        clip(xg).square().mean().backward()
    print(f"MAPBlock B={B:5d}: forward {f:6.3f} ms, forward+backward {timed(fb):6.3f} ms", flush=True)
ls = torch.tensor(2.659, device="cuda", requires_grad=True)
for B in (128, 1024, 2048):
    a, b = torch.randn(B, 384, device="cuda", requires_grad=True), torch.randn(B, 384, device="cuda", requires_grad=True)
    def step():
        a.grad = b.grad = ls.grad = None
        clip_auxiliary_loss(a, b, ls).backward()
    def eager():
        import torch.nn.functional as F
        a.grad = b.grad = ls.grad = None
        i, l = F.normalize(a, dim=-1), F.normalize(b, dim=-1)
        s = ls.exp() * i @ l.t()
        lab = torch.arange(B, device="cuda")
        ((F.cross_entropy(s, lab) + F.cross_entropy(ls.exp() * l @ i.t(), lab)) / 2).backward()
    print(f"InfoNCE  B={B:5d}: HIP value+gradients {timed(step):6.3f} ms   (PyTorch-ROCm eager, same GPU: {timed(eager):6.3f} ms)", flush=True)
