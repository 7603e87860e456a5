"""Where the host time of a rollout-sized (B = 1) sample_ddim call goes: cProfile over 300 graph-replayed calls (no synchronisation
between them) and the per-call wall time of the host side alone."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mdt_policy_amd import synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
dev = torch.device("cuda")
cfg, P, model = bench.build_model(dev)
sig = gs.get_sigmas_exponential(10, 0.001, 80.0)
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(1, cfg, seed=1).items()}
st = {"state_images": inp["state_images"], "modality": "lang"}
x = inp["noise"] * 80
with torch.no_grad():
    for _ in range(10): gs.sample_ddim(model, st, x, inp["goal"], sig)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): gs.sample_ddim(model, st, x, inp["goal"], sig)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host side of a call: {(t1 - t0) / n * 1e6:.1f} us (queue drained {1e3 * (t2 - t1):.1f} ms later)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n): gs.sample_ddim(model, st, x, inp["goal"], sig)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
