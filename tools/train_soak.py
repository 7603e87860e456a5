#!/usr/bin/env python3
"""Determinism soak of the training path with its side stream (round 6): the same seeded step N times -- forward, HIP backward with the
weight gradients / decoder tail beside the chain, the forward's preparation beside the encoder -- must leave the SAME BITS in the loss
and in every gradient every time (every kernel is deterministic; a difference is a race between the streams), at several batch sizes,
in eval and train mode (the dropout masks are a function of the seed).  usage: python tools/train_soak.py [iterations=200] [batches...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import configs, synthetic
from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
batches = [int(a) for a in sys.argv[2:]] or [1024, 128, 37]
cfg = configs.mdtv_default()
bad = 0
for B in batches:
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda()
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"].requires_grad_(), "modality": "lang"}
    goal = inp["goal"].requires_grad_()
    for mode in ("eval", "train"):
        model.train(mode == "train")
        ref = None
        for it in range(N):
            torch.manual_seed(1234)  # the dropout seed of the step comes from torch's CPU generator
            model.zero_grad(set_to_none=True)
            state["state_images"].grad = None
            goal.grad = None
            loss, _ = model.loss(state, li["actions"], goal, li["noise_train"], li["sigma"])
            (loss + 0.1 * model.inner_model.latent_encoder_emb.square().mean()).backward()
            got = [loss.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None] + \
                  [state["state_images"].grad.clone(), goal.grad.clone()]
            if ref is None:
                ref = got
            elif not all(torch.equal(a, b) for a, b in zip(ref, got)):
                bad += 1
                print(f"MISMATCH B={B} {mode} iteration {it}: {[i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b)][:8]}", flush=True)
        torch.cuda.synchronize()
        print(f"B={B:5d} {mode:5s}: {N} identical steps compared, mismatching steps so far {bad}", flush=True)
assert bad == 0, f"{bad} steps differed from the first one"
print("train soak ok")
