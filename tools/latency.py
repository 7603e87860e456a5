"""Sampler latency / throughput vs batch size (eager launches on the current stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mdt_policy_amd import synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
dev = torch.device("cuda")
cfg, P, model = bench.build_model(dev)
sig = gs.get_sigmas_exponential(10, 0.001, 80.0)
if os.environ.get("LAT_DEV_SIG") == "1": sig = sig.to(dev)  # the schedule on the device, as the agent builds it (no copy per call)
for B in [int(x) for x in (sys.argv[1:] or ["1", "4", "16", "64", "256", "1024"])]:
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, seed=1).items()}
    st = {"state_images": inp["state_images"], "modality": "lang"}
    x = inp["noise"] * 80
    with torch.no_grad():
        for _ in range(5): gs.sample_ddim(model, st, x, inp["goal"], sig)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
        for _ in range(n): out = gs.sample_ddim(model, st, x, inp["goal"], sig)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        t1 = time.perf_counter()
        for _ in range(n): out = gs.sample_ddim(model, st, x, inp["goal"], sig); torch.cuda.synchronize()
        dl = (time.perf_counter() - t1) / n
    print(f"B={B:5d}: {dt*1e3:8.3f} ms/call pipelined ({B/dt:9.0f} chunks/s), {dl*1e3:8.3f} ms/call synchronous latency", flush=True)
