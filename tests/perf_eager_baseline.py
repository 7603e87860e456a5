#!/usr/bin/env python3
"""Measurement aid (not collected by pytest): the reference ALGORITHM as plain PyTorch-ROCm eager on the same MI355X
-- the oracle's functions fed with GPU tensors, fp32 -- next to the HIP path, for sampling (B=256 and B=1, 10 DDIM
steps) and for one training step (B=128, loss forward + autograd backward + AdamW).  This is what a user of the
reference gets on this GPU without this repository.   usage: python tests/perf_eager_baseline.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import configs, synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
from oracle import mdt_oracle as O

dev = torch.device("cuda")
torch.backends.cuda.matmul.allow_tf32 = False
cfg = configs.mdtv_default()
model = GCDenoiser(cfg, 0.5)
shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
P = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.fill_state_dict(shapes, seed=0, profile="init").items()}
model.load_state_dict(P)
model = model.to(dev).eval()
sig = O.get_sigmas_exponential(10, 0.001, 80.0)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in (256, 1):
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    st = {"state_images": inp["state_images"], "modality": "lang"}
    x = inp["noise"] * 80.0
    with torch.no_grad():
        t_eager = timed(lambda: O.sample_ddim(P, cfg, st, x, inp["goal"], sig.to(dev), hoist=False), 5)
        t_hip = timed(lambda: gs.sample_ddim(model, st, x, inp["goal"], sig), 20)
    print(f"sampling B={B:3d}: torch eager {t_eager * 1e3:8.2f} ms ({B / t_eager:9.0f} chunks/s)   "
          f"HIP {t_hip * 1e3:7.2f} ms ({B / t_hip:9.0f} chunks/s)   x{t_eager / t_hip:.1f}", flush=True)

TRAIN_B = [int(v) for v in os.environ.get("MDT_EAGER_TRAIN_B", "128").split(",")]
B = TRAIN_B[0]
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
li = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
st = {"state_images": inp["state_images"], "modality": "lang"}
Pg = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in P.items()}
used = [k for k in Pg if "proprio_emb" not in k and "rotary" not in k and k != "inner_model.pos_emb" and "goal_emb" not in k]
opt_e = torch.optim.AdamW([Pg[k] for k in used], lr=1e-4, fused=True)
opt_h = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)


def step_eager():
    opt_e.zero_grad(set_to_none=True)
    loss, _ = O.loss(Pg, cfg, st, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
    loss.backward()
    opt_e.step()


def step_hip():
    opt_h.zero_grad(set_to_none=True)
    loss, _ = model.loss(st, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
    loss.backward()
    opt_h.step()


te, th = timed(step_eager, 10), timed(step_hip, 20)
print(f"training B={B}: torch eager {te * 1e3:8.2f} ms/step ({B / te:8.0f} samples/s)   HIP {th * 1e3:7.2f} ms/step "
      f"({B / th:8.0f} samples/s)   x{te / th:.1f}")

# ---- Perceiver resampler, forward + backward at the training batch
from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
from oracle import perceiver_oracle as PO

res = PerceiverResampler(dim=384, depth=6, dim_head=64, heads=8, num_latents=3, num_time_embeds=1).to(dev)
Pp = {k: v.detach().clone().requires_grad_() for k, v in res.state_dict().items()}
xm = torch.randn(128, 1, 392, 384, device=dev)
cot = torch.randn(128, 3, 384, device=dev)


def perc_eager():
    for v in Pp.values():
        v.grad = None
    (PO.perceiver_resampler(Pp, xm, 8) * cot).sum().backward()


def perc_hip():
    res.zero_grad(set_to_none=True)
    (res(xm) * cot).sum().backward()


te, th = timed(perc_eager, 5), timed(perc_hip, 5)
print(f"perceiver fwd+bwd B=128: torch eager {te * 1e3:8.2f} ms   HIP {th * 1e3:7.2f} ms   x{te / th:.1f}")
with torch.no_grad():
    te, th = timed(lambda: PO.perceiver_resampler(Pp, xm, 8), 5), timed(lambda: res(xm), 10)
print(f"perceiver fwd     B=128: torch eager {te * 1e3:8.2f} ms   HIP {th * 1e3:7.2f} ms   x{te / th:.1f}")
