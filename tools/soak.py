"""Soak: many sampler calls (B=1 and B=64) and training steps; device / host memory must stay flat."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd import configs, synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser

dev = torch.device("cuda")
cfg = configs.mdtv_default()
model = GCDenoiser(cfg, 0.5).to(dev).eval()
sig = gs.get_sigmas_exponential(10, 0.001, 80.0)


def mem():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2 ** 20, torch.cuda.memory_allocated() / 2 ** 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024


def phase(name, fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    m0 = mem()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    m1 = mem()
    print(f"{name:28s} x{n}: device used {m0[0]:.0f} -> {m1[0]:.0f} MiB, torch {m0[1]:.1f} -> {m1[1]:.1f} MiB, host RSS {m0[2]:.0f} -> {m1[2]:.0f} MiB", flush=True)
    assert m1[0] - m0[0] < 64 and m1[2] - m0[2] < 64, "memory grew"


for B in (1, 64):
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    st = {"state_images": inp["state_images"], "modality": "lang"}
    with torch.no_grad():
        phase(f"sample_ddim B={B}", lambda: gs.sample_ddim(model, st, inp["noise"] * 80.0, inp["goal"], sig), 1000)
        phase(f"sample_heun B={B}", lambda: gs.sample_heun(model, st, inp["noise"] * 80.0, inp["goal"], sig), 100)
B = 128
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
li = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
st = {"state_images": inp["state_images"], "modality": "lang"}
from mdt_policy_amd.models.contrastive import compute_contrastive_loss
from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
from mdt_policy_amd.optim import FusedAdamW

clip = ClipStyleProjection("map", 384, 1, 4).to(dev)
logit_scale = torch.nn.Parameter(torch.tensor(2.659, device=dev))
opt = FusedAdamW(list(model.parameters()) + list(clip.parameters()) + [logit_scale], lr=1e-5)
model.train()


def step():
    """The agent's training step: diffusion loss (language goal) + contrastive loss against the vision-goal context
    (HIP denoiser forward / backward, HIP MAPBlock, HIP InfoNCE), FusedAdamW."""
    opt.zero_grad(set_to_none=True)
    loss, _ = model.loss(st, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
    cont = compute_contrastive_loss(model, clip, logit_scale, dict(st, modality="vis"), inp["goal"], li["actions"],
                                    li["sigma"], li["noise_train"])
    (loss + cont).backward()
    opt.step()


phase("training step B=128", step, 300)
# forwards whose graphs are dropped without a backward must hand their tapes back
phase("loss without backward", lambda: model.loss(st, li["actions"], inp["goal"], li["noise_train"], li["sigma"]), 300)
print("soak ok")
