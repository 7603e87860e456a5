import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd import configs, synthetic
from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
from mdt_policy_amd.optim import FusedAdamW
B = 1024
cfg = configs.mdtv_default()
for which in ("torch", "fused", "none"):
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda().train()
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True) if which == "torch" else FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        if which != "none": opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); print(which, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms/step", flush=True)
    # optimizer alone
    if which != "none":
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): opt.step()
        torch.cuda.synchronize(); print("   opt.step alone", round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms", flush=True)
