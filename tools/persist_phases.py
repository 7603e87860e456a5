"""Where does the persistent decoder kernel spend its time?  Per phase kind: body time of the workgroups that held a tile
(mean / max over workgroups), and the phase span (start of this phase -> start of the next = body + barrier wait), from
the shader-clock stamps of mdt_persist_set_debug.   usage: python tools/persist_phases.py [B ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mdt_policy_amd import _lib, synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs

NAMES = {0: "qkv GEMM (LN+mod)", 1: "fc GEMM (LN+mod, GELU)", 2: "proj GEMM (+res)", 102: "c_proj GEMM K=4d (+res)", 3: "small qkv", 103: "small fc",
         203: "small c_proj", 4: "self-attention", 5: "attention+proj", 6: "cross-attention (folded)", 7: "head"}
dev = torch.device("cuda")
cfg, P, model = bench.build_model(dev)
lib = _lib.load()
lib.mdt_persist_set_debug.restype = C.c_int32
lib.mdt_persist_set_debug.argtypes = [C.c_void_p, C.c_void_p]
lib.mdt_persist_phase_count.restype = C.c_int32
lib.mdt_persist_phase_count.argtypes = [C.c_void_p]
lib.mdt_persist_phase_kind.restype = C.c_int32
lib.mdt_persist_phase_kind.argtypes = [C.c_int32] * 4
sig = gs.get_sigmas_exponential(10, 0.001, 80.0)
eng = model.inner_model.hip_engine(0.5)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
for B in [int(x) for x in (sys.argv[1:] or ["256", "1"])]:
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.sampler_inputs(B, cfg, seed=1).items()}
    st = {"state_images": inp["state_images"], "modality": "lang"}
    x = inp["noise"] * 80
    with torch.no_grad():
        for _ in range(3):
            gs.sample_ddim(model, st, x, inp["goal"], sig)
        torch.cuda.synchronize()
        nph = lib.mdt_persist_phase_count(eng.handle)
        if nph == 0:
            print(f"B={B}: the persistent kernel is not used"); continue
        buf = torch.zeros(ncu * nph * 2 + ncu, dtype=torch.int64, device=dev)
        lib.mdt_persist_set_debug(eng.handle, buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gs.sample_ddim(model, st, x, inp["goal"], sig); e1.record()
        torch.cuda.synchronize()
        lib.mdt_persist_set_debug(eng.handle, None)
    raw = buf.cpu().numpy()
    t = raw[: ncu * nph * 2].reshape(ncu, nph, 2)
    start = t[:, :, 0]
    busy = (t[:, :, 1] < 0)
    end = t[:, :, 1] & 0x7FFFFFFFFFFFFFFF
    live = start[:, 0] != 0          # workgroups of XCDs that own samples
    small = B <= 8
    kinds = np.array([lib.mdt_persist_phase_kind(0, i, 4, int(small)) for i in range(nph)])
    kern = (end[live][:, -1].max() - start[live][:, 0].min())
    ms = e0.elapsed_time(e1)
    print(f"B={B}: call {ms:.3f} ms (with stamps); kernel span {kern} cycles, {nph} phases, {int(live.sum())} live workgroups")
    body = (end - start).astype(np.float64)
    span = np.diff(start, axis=1).astype(np.float64)  # phase p start -> phase p+1 start
    print(f"  {'phase':28s} {'count':>5s} {'body mean':>10s} {'body max':>10s} {'span':>10s} {'sum span':>12s}   (cycles)")
    tot = 0.0
    for k in sorted(set(kinds.tolist())):
        idx = np.where(kinds == k)[0]
        idx2 = idx[idx < nph - 1]
        bm = np.array([body[live & busy[:, i], i].mean() for i in idx if (live & busy[:, i]).any()])
        bx = np.array([body[live & busy[:, i], i].max() for i in idx if (live & busy[:, i]).any()])
        sp = np.array([span[live, i].mean() for i in idx2])
        tot += sp.sum()
        print(f"  {NAMES.get(k, str(k)):28s} {len(idx):5d} {bm.mean():10.0f} {bx.mean():10.0f} {sp.mean():10.0f} {sp.sum():12.0f}")
    print(f"  sum of spans {tot:.0f} cycles = {tot / kern * 100:.1f} % of the kernel span; cycles per ms ~ {kern / ms:.0f}")
