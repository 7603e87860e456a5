"""Can one fused sampler call be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed?  B = 1 rollout latency with and
without.  usage: python tools/graph_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mdt_policy_amd import synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs

dev = torch.device("cuda")
cfg, P, model = bench.build_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, seed=1).items()}
st = {"state_images": inp["state_images"], "modality": "lang"}
x = inp["noise"] * 80
sig = gs.get_sigmas_exponential(10, 0.001, 80.0).to(dev)
with torch.no_grad():
    for _ in range(5):
        ref = gs.sample_ddim(model, st, x, inp["goal"], sig)
    torch.cuda.synchronize()

    def lat(fn, n=200):
        for _ in range(10): fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    print(f"B={B}: eager   {lat(lambda: gs.sample_ddim(model, st, x, inp['goal'], sig)):.3f} ms per synchronised call")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            gs.sample_ddim(model, st, x, inp["goal"], sig)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            out = gs.sample_ddim(model, st, x, inp["goal"], sig)
    except Exception as e:
        print("capture failed:", repr(e)[:400]); sys.exit(0)
    graph.replay(); torch.cuda.synchronize()
    print("replay equals eager:", bool(torch.equal(out, ref)))
    print(f"B={B}: replay  {lat(lambda: graph.replay()):.3f} ms per synchronised call")
