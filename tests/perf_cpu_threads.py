"""Time the CPU oracle at several thread counts on this host (sizes bench.py's cpu_baseline default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd import configs, synthetic
from oracle import mdt_oracle as O
cfg = configs.mdtv_default()
import json
man = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "state_dict_manifest.json")))
shapes = [(k, tuple(s)) for k, s in man["mdtv_default"]["state_dict"]]
P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 0, "init").items()}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inp = {k: torch.from_numpy(v) for k, v in synthetic.sampler_inputs(B, cfg, seed=1).items()}
st = {"state_images": inp["state_images"], "modality": "lang"}
sig = O.get_sigmas_exponential(10, 0.001, 80.0)
for th in (1, 8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    O.sample_ddim(P, cfg, {"state_images": inp["state_images"][:8], "modality": "lang"}, inp["noise"][:8], inp["goal"][:8], sig)
    t0 = time.perf_counter(); O.sample_ddim(P, cfg, st, inp["noise"] * 80, inp["goal"], sig, hoist=False); dt = time.perf_counter() - t0
    print(f"threads {th:4d}: B={B} {dt:.3f}s -> {B/dt:.1f} chunks/s", flush=True)
    if dt > 30: break
