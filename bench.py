#!/usr/bin/env python3
"""bench.py -- action-chunks/sec of the MDT denoising hot path on MI355X (BASELINE.json metric).

A "step" is ONE full sampler invocation over one batch: encoder + cross-K/V once, then 10 DDIM denoise steps of the
MDT-V d=384 4+4-block score network over B=256 action chunks (horizon 10, action dim 7) per GPU -- BASELINE config
C2 at N=1, C4-shaped (256 per GPU, weak scaling, one RCCL all-gather of the sampled actions per step) at N>1.
Inputs (state tokens, goal, x_T) and the random-init weights are synthetic and already resident in HBM when the
timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  roofline     : algorithmic FLOPs (mdt_flops_per_chunk = 1.812 GFLOP/chunk) / HIP-event time of the timed region
                 against the dense FP32-MFMA peak (157.3 TFLOP/s; bf16/fp16 operands fail the parity gate), plus
                 the same figure for the dominant kernel (the fused MFMA GEMM) timed alone through the op-level ABI
  cpu_baseline : the CPU oracle ("port" of the reference algorithm, as-written: encoder re-run every step) timed on
                 this host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X dense FP32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense BF16 matrix peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
SPEC_MHZ = 2400.0              # the shader clock the FP32-matrix peak is quoted at (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)
T_START = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="action chunks per GPU per step")
    ap.add_argument("--denoise-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the CPU baseline (all legs together)")
    ap.add_argument("--cpu-leg", default="", help=argparse.SUPPRESS)  # internal: "threads,hoist,budget" -> one CPU-baseline leg, JSON on stdout
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: min(cpu_count, 16), the best of a 1..128 sweep on the EPYC host)")
    return ap.parse_args()


def build_model(device):
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    cfg = configs.mdtv_default()
    model = GCDenoiser(cfg, sigma_data=0.5)
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=0, profile="init").items()}
    model.load_state_dict(P)
    return cfg, P, model.to(device).eval()


def sustained_mhz(stamps):
    """Median over the XCDs of (d s_memtime / d s_memrealtime) x 100 MHz between two mdt_op_clock_stamp records (2, 8, 2)."""
    st = stamps.cpu().view(2, 8, 2)
    vals = []
    for x in range(8):
        if int(st[0, x, 1]) and int(st[1, x, 1]):
            d_clk, d_ref = int(st[1, x, 0] - st[0, x, 0]), int(st[1, x, 1] - st[0, x, 1])
            if d_clk > 0 and d_ref > 0:
                vals.append((d_clk / d_ref * 100.0, d_ref / 100.0))
    if not vals:
        return None, None
    vals.sort()
    return round(vals[len(vals) // 2][0], 1), vals[len(vals) // 2][1]


def time_dominant_kernel(device, M):
    """The dominant kernel of the path: the fused MLP sublayer k_mlp (LayerNorm + adaLN-modulate prologue -> c_fc (N = 1536,
    K = 384) -> GELU -> c_proj (N = 384, K = 1536) -> gate, one launch, 40 launches per sampler call, a third of its FLOPs),
    timed alone through the op-level C ABI with HIP events on the launch stream."""
    from mdt_policy_amd import _lib
    lib = _lib.load()
    D, N = 384, 1536
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, D, generator=g).to(device)
    W1 = (torch.randn(N, D, generator=g) * 0.02).to(device)
    W2 = (torch.randn(D, N, generator=g) * 0.02).to(device)
    lw = torch.ones(D, device=device)
    mod = torch.randn(6 * D, generator=g).to(device)
    P1, P2 = torch.zeros(N * D, device=device), torch.zeros(N * D, device=device)
    S = 4 * D // 512
    parts = torch.empty(S, M, D, device=device)
    s = torch.cuda.current_stream(device).cuda_stream
    _lib.check(lib.mdt_op_pack_weight(W1.data_ptr(), N, D, P1.data_ptr(), 0, N, s))
    _lib.check(lib.mdt_op_pack_weight(W2.data_ptr(), D, N, P2.data_ptr(), 0, D, s))
    f, p = _lib.GemmArgs(), _lib.GemmArgs()
    f.A, f.lda, f.Wp, f.M, f.N, f.K = A.data_ptr(), D, P1.data_ptr(), M, N, D
    p.A, p.lda, p.Wp, p.M, p.N, p.K, p.ldo = A.data_ptr(), D, P2.data_ptr(), M, D, N, D
    f.ln, f.ln_w, f.act = 1, lw.data_ptr(), _lib.ACT["gelu"]
    f.mod = p.mod = mod.data_ptr()
    f.mod_stride = p.mod_stride = 0
    f.shift_off, f.scale_off, f.gate_off, p.shift_off, p.scale_off, p.gate_off = 3 * D, 4 * D, -1, -1, -1, 5 * D
    for a in (f, p):
        a.rows_per_sample, a.gin, a.gout, a.goff = 10, 1, 1, 0
    n = C.c_int32(0)
    # the form the model-level calls use: the three-way bf16 split (round 6) unless MDT_HIP_MLP_SPLIT=0
    split = os.environ.get("MDT_HIP_MLP_SPLIT", "1") != "0"
    if split:
        S1 = torch.zeros(N * D * 6, dtype=torch.uint8, device=device)
        S2 = torch.zeros(N * D * 6, dtype=torch.uint8, device=device)
        _lib.check(lib.mdt_op_pack_weight_split(W1.data_ptr(), N, D, S1.data_ptr(), s))
        _lib.check(lib.mdt_op_pack_weight_split(W2.data_ptr(), D, N, S2.data_ptr(), s))

    def launch():
        if split:
            _lib.check(lib.mdt_op_mlp_split(C.byref(f), C.byref(p), S1.data_ptr(), S2.data_ptr(), parts.data_ptr(), M * D, C.byref(n), s))
        else:
            _lib.check(lib.mdt_op_mlp(C.byref(f), C.byref(p), parts.data_ptr(), M * D, C.byref(n), s))
    for _ in range(10):
        launch()
    reps = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * M * N * D * 2
    tf = flops / (us * 1e-6) / 1e12
    # bytes of one launch: TRUE algorithmic = rows read once, both weight images once, the sublayer's output written once;
    # the kernel as built writes S partial slabs instead of one output (an implementation artefact, summed by the next reader)
    wbytes = 6 if split else 4   # the split images hold three bf16 parts per weight
    alg_true = 4 * (M * D + M * D) + wbytes * 2 * N * D
    alg_slabs = 4 * (M * D + S * M * D) + wbytes * 2 * N * D
    form = ("three-way bf16 split: every fp32 operand as three bf16 parts, six v_mfma_f32_16x16x32_bf16 products per k32 step, fp32 "
            "accumulation (fp32 product accuracy; 6 executed bf16 FLOPs per algorithmic FLOP: `frac` = algorithmic FLOPs against the "
            "FP32 matrix peak, `frac_of_bf16_peak_executed` = executed bf16 MFMA FLOPs against the 2.5 PFLOP/s dense bf16 peak)"
            if split else "v_mfma_f32_16x16x4_f32")
    return {"name": f"{'k_mlp_split' if split else 'k_mlp'} LN+mod -> c_fc -> GELU -> c_proj -> gate ({M} rows, d={D}, hidden {N}: "
                    f"{flops / 1e9:.2f} GFLOP per launch)",
            "form": form, "executed_flops_per_algorithmic": 6 if split else 1,
            "gflop_per_launch": round(flops / 1e9, 3),
            "alone": {"avg_us": round(us, 2), "achieved": round(tf, 2), "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                      "what": f"{reps} back-to-back launches of the kernel alone through the op-level C ABI (HIP events on the launch "
                              "stream); reads slower than the kernel inside its chain: with nothing else in between, a launch's "
                              "workgroups wait for the late waves of the previous one"},
            "algorithmic_bytes_per_launch": alg_true,
            "algorithmic_bytes_per_launch_with_slabs": alg_slabs}


def time_dominant_kernel_in_chain(step, device):
    """The same kernel INSIDE the sampler call: every k_mlp launch of one more call between its own pair of HIP events on the
    launch stream (mdt_op_trace_mlp) -- what a kernel trace reports for it (profiles/r04_bench_kernel_stats.txt)."""
    from mdt_policy_amd import _lib
    lib = _lib.load()
    step()
    torch.cuda.synchronize(device)
    lib.mdt_op_trace_mlp(1)
    try:
        step()
        torch.cuda.synchronize(device)
    finally:
        lib.mdt_op_trace_mlp(0)
    buf = (C.c_float * 4096)()
    n = lib.mdt_op_trace_mlp_read(buf, 4096)
    raw = [buf[i] for i in range(n)]
    if not raw:
        return None
    ebuf = (C.c_float * 4096)()
    ne = lib.mdt_op_trace_mlp_read_empty(ebuf, 4096)
    empty = sorted(ebuf[i] for i in range(ne))
    # an event pair on the stream costs something by itself (the record packets and the boundary between them): behind every
    # traced launch the hook brackets NOTHING with a second pair; their median is subtracted from every launch bracket
    e_med = empty[len(empty) // 2] if empty else 0.0
    us = sorted(max(0.0, v - e_med) for v in raw)
    return {"launches": n, "avg_us": round(sum(us) / n, 2), "median_us": round(us[n // 2], 2), "min_us": round(us[0], 2),
            "max_us": round(us[-1], 2), "avg_us_bracket": round(sum(raw) / n, 2), "empty_bracket_us": round(e_med, 2),
            "what": "launch bracket minus the median EMPTY bracket recorded behind each launch (mdt_op_trace_mlp_read_empty)"}


def other_configs(device):
    """Diagnostic numbers for BASELINE.json's other single-GPU configurations (not the metric): the MDT-V training step
    at B = 1024 (configs[2]: diffusion loss forward + backward + AdamW, train() mode with the shipped dropout rates) and
    the rollout batch B = 1 (10 DDIM steps)."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.optim import FusedAdamW
    out = {}
    cfg = configs.mdtv_default()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).to(device)

    from mdt_policy_amd import _lib as _L
    stamps = torch.zeros(2, 16, dtype=torch.int64, device=device)
    last_mhz = [None]

    def timed(fn, warm, n):
        """Mean seconds per call of n calls behind `warm` warm-ups; last_mhz[0]: the shader clock sustained over the timed calls
        (mdt_op_clock_stamp in front of and behind them on the launch stream)."""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(device)
        ls = torch.cuda.current_stream(device).cuda_stream
        stamps.zero_()
        t0 = time.perf_counter()
        _L.check(_L.load().mdt_op_clock_stamp(stamps[0].data_ptr(), ls))
        for _ in range(n):
            fn()
        _L.check(_L.load().mdt_op_clock_stamp(stamps[1].data_ptr(), ls))
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / n
        last_mhz[0] = sustained_mhz(stamps)[0]
        return dt

    B = 1024
    inp = {k: torch.from_numpy(v).to(device) for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).to(device) for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    model.train()

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        opt.step()

    dt = timed(step, 3, 10)
    # algorithmic FLOPs of a training step: forward as written (one sigma per sample: the encoder cannot be hoisted,
    # 242.8 MFLOP / sample = mdt_flops_per_chunk(1)) + backward ~ 2x forward
    fwd = model.inner_model.hip_engine(0.5).flops_per_chunk(1)
    tf = 3.0 * fwd * B / dt / 1e12
    out["train_step_mdtv_B1024"] = {"ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1), "sustained_mhz": last_mhz[0],
                                    "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                                                 "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                                                 "flops_per_sample": round(3.0 * fwd)},
                                    "what": "GCDenoiser.loss forward + HIP backward + FusedAdamW, train() mode, dropout "
                                            "0.3/0.1/0.05, fp32, synthetic CALVIN-shaped batch (diffusion loss only; the full "
                                            "configs[2] step is train_step_c3_mdtv_B1024)"}
    del opt
    # BASELINE configs[2] in full: diffusion loss + the masked-token auxiliary (masked generative foresight head on
    # latent_encoder_emb, mdtv_agent.py:258-269, masked_beta = 1) in ONE optimizer step over both modules
    try:
        from mdt_policy_amd.models.img_generation.masked_transformer_decoder import MaskedTransformerImgDecoder
        from tools.mae_bench import mae_flops
        gen = MaskedTransformerImgDecoder(resolution=112, patch_size=16, decoder_depth=6, decoder_embed_dim=192, decoder_n_heads=8,
                                          context_dim=384, mask_ratio=0.75).to(device)
        imgs = torch.randn(B, 2, 3, 112, 112, device=device)
        opt2 = FusedAdamW(list(model.parameters()) + list(gen.parameters()), lr=1e-4, weight_decay=0.05)

        def step_c3():
            opt2.zero_grad(set_to_none=True)
            loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
            rec, mask, restore, _ = gen(model.inner_model.latent_encoder_emb, imgs)
            (loss + gen.compute_loss(imgs, rec, mask, restore)).backward()
            opt2.step()

        dt3 = timed(step_c3, 2, 5)
        tf3 = (3.0 * fwd * B + mae_flops(B)) / dt3 / 1e12
        out["train_step_c3_mdtv_B1024"] = {"ms_per_step": round(dt3 * 1e3, 3), "samples_per_s": round(B / dt3, 1), "sustained_mhz": last_mhz[0],
                                           "roofline": {"bound": "mfma", "achieved": round(tf3, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                                                        "unit": "TFLOP/s", "frac": round(tf3 / PEAK_FP32_MFMA_TFLOPS, 4),
                                                        "flops_per_sample": round(3.0 * fwd + mae_flops(1))},
                                           "what": "BASELINE configs[2]: diffusion loss + masked-token aux (MaskedTransformerImgDecoder, 6 "
                                                   "blocks d=192, 112x112 frames, mask 0.75) forward + HIP backward + FusedAdamW, fp32"}
        del opt2, gen, imgs
    except Exception as e:  # diagnostic leg
        out["train_step_c3_mdtv_B1024"] = {"error": repr(e)}
    model.eval()
    sig = gs.get_sigmas_exponential(10, 0.001, 80.0).to(device)  # ON the device, as MDTVAgent.get_noise_schedule builds it
    # weight-streaming regime: a call reads the decoder's weights once per step and the encoder's once
    P = dict(model.state_dict())
    enc_b = 4 * sum(v.numel() for k, v in P.items() if k.startswith(("inner_model.encoder.", "inner_model.tok_emb", "inner_model.lang_emb",
                                                                    "inner_model.goal_emb")) and "proprio" not in k)
    dec_b = 4 * sum(v.numel() for k, v in P.items() if k.startswith(("inner_model.decoder.", "inner_model.sigma_emb", "inner_model.action_")))
    low = {}
    for b in (1, 16):
        one = {k: v[:b].contiguous() for k, v in inp.items()}
        stb = {"state_images": one["state_images"], "modality": "lang"}
        xT = one["noise"] * 80.0
        with torch.no_grad():
            dt_p = timed(lambda: gs.sample_ddim(model, stb, xT, one["goal"], sig), 5, 50)

            def sync_call():
                gs.sample_ddim(model, stb, xT, one["goal"], sig)
                torch.cuda.synchronize(device)
            dt_s = timed(sync_call, 5, 50)
        byts = dec_b * 10 + enc_b
        low[f"B{b}"] = {"ms_per_call_pipelined": round(dt_p * 1e3, 3), "ms_per_call_synchronous": round(dt_s * 1e3, 3),
                        "bytes_per_call": byts, "achieved": round(byts / dt_s / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(byts / dt_s / 1e9 / PEAK_HBM_GBS, 4)}
    out["rollout_B1_10steps"] = {"ms_per_chunk": low["B1"]["ms_per_call_synchronous"],
                                 "ms_per_chunk_pipelined": low["B1"]["ms_per_call_pipelined"],
                                 "what": "one sample_ddim call for ONE chunk, sigmas on the device as the agent passes them, "
                                         "host-synchronised after every call (rollout latency)"}
    out["roofline_lowbatch"] = {"bound": "hbm", "bytes": f"decoder weights {dec_b} B x 10 steps + encoder weights {enc_b} B once "
                                                          "(SURVEY.md 8(d); activations negligible)", **low}
    # BASELINE configs[4] stand-in (the real one needs CALVIN, the simulator and the Voltron / CLIP weights): what MDTVAgent.forward
    # runs per replan behind its frozen encoders, composed as the agent composes it (mdtv_agent.py:392-403, 688-719) --
    # Voltron-shaped patch tokens of both cameras (1, 1, 2 x 196, 384) -> PerceiverResampler (6 layers, 3 latents) -> state_images
    # (1, 3, 384) + CLIP-shaped language goal (1, 1, 512) -> x_T = randn * sigma_max -> sample_ddim, host-synchronised per call
    try:
        from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
        perc = PerceiverResampler(dim=384, depth=6, dim_head=64, heads=8, num_latents=3, num_time_embeds=1).to(device).eval()
        tokens = torch.randn(1, 2 * 196, 384, device=device)
        goal1 = torch.randn(1, 512, device=device)

        def replan():
            perceptual_emb = {"state_images": perc(tokens.unsqueeze(1)), "modality": "lang"}
            latent_goal = goal1.unsqueeze(1)
            x = torch.randn((1, 10, 7), device=device) * 80.0
            act = gs.sample_ddim(model, perceptual_emb, x, latent_goal, sig)
            torch.cuda.synchronize(device)
            return act

        with torch.no_grad():
            dt_r = timed(replan, 8, 50)

            def perc_only():
                perc(tokens.unsqueeze(1))
                torch.cuda.synchronize(device)
            dt_pc = timed(perc_only, 5, 50)
        out["rollout_e2e_synthetic_B1"] = {"ms_per_replan": round(dt_r * 1e3, 3), "ms_perceiver_alone": round(dt_pc * 1e3, 3),
                                           "what": "BASELINE configs[4] stand-in: synthetic Voltron-shaped tokens (1, 1, 392, 384) -> HIP "
                                                   "PerceiverResampler -> (1, 3, 384) + goal (1, 1, 512) -> randn x_T -> sample_ddim "
                                                   "(10 steps), composed as MDTVAgent.forward does, host-synchronised per replan; the "
                                                   "frozen image / language encoders in front of it are not part of this repository"}
    except Exception as e:  # diagnostic leg
        out["rollout_e2e_synthetic_B1"] = {"error": repr(e)}
    # the other samplers of the agent's dispatch table at the metric's own batch: 10 sigma steps each
    try:
        B2 = 256
        st2 = {"state_images": inp["state_images"][:B2].contiguous(), "modality": "lang"}
        g2, x2 = inp["goal"][:B2].contiguous(), inp["noise"][:B2].contiguous() * 80.0
        sig_h = gs.get_sigmas_exponential(10, 0.001, 80.0)
        for name, fn, evals in (("sample_heun", gs.sample_heun, 19), ("sample_dpmpp_2m", gs.sample_dpmpp_2m, 10), ("sample_euler", gs.sample_euler, 10)):
            with torch.no_grad():
                dt_s = timed(lambda: fn(model, st2, x2, g2, sig_h), 3, 10)
            out[f"{name}_B256"] = {"ms_per_call": round(dt_s * 1e3, 3), "chunks_per_s": round(B2 / dt_s, 1), "model_evals": evals,
                                   "what": f"gc_sampling.{name} over 10 exponential sigma steps, B = 256, encoder hoisted, one denoiser "
                                           "evaluation per launch sequence (host-driven step loop)"}
    except Exception as e:  # diagnostic leg
        out["other_samplers_B256"] = {"error": repr(e)}
    # the other shipped architecture: MDTTransformer (d = 512, 4 + 6 blocks; configs.mdt_default), the same fused sampler call at B = 256
    try:
        from mdt_policy_amd import configs as _cfgs
        from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser as _GCD
        cfg_m = _cfgs.mdt_default()
        torch.manual_seed(0)
        m_mdt = _GCD(cfg_m, 0.5).to(device).eval()
        inp_m = {k: torch.from_numpy(v).to(device) for k, v in synthetic.sampler_inputs(256, cfg_m, 3, "mdt").items()}
        st_m = {"static": inp_m["static"], "gripper": inp_m["gripper"], "modality": "lang"}
        sig_m = gs.get_sigmas_exponential(10, 0.001, 80.0).to(device)
        with torch.no_grad():
            dt_m = timed(lambda: m_mdt.sample_ddim(st_m, inp_m["noise"] * 80.0, inp_m["goal"], sig_m), 3, 10)
        out["sample_ddim_mdt_arch_B256"] = {"ms_per_call": round(dt_m * 1e3, 3), "chunks_per_s": round(256 / dt_m, 1),
                                            "what": "MDTTransformer (d = 512, 4 encoder + 6 decoder blocks, configs.mdt_default), fused sample_ddim, "
                                                    "10 steps, B = 256, random-init weights"}
        del m_mdt
    except Exception as e:  # diagnostic leg
        out["sample_ddim_mdt_arch_B256"] = {"error": repr(e)}
    # the contrastive (CLA) auxiliary head at the training batch (mdtv_agent.py:440-484): the MAP pooling
    # block over latent_encoder_emb, forward + backward, and the InfoNCE op (value + gradients)
    try:
        from mdt_policy_amd.models.contrastive import clip_auxiliary_loss
        from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
        clip = ClipStyleProjection("map", 384, 1, 4).to(device)
        xg = torch.randn(1024, 4, 384, device=device, requires_grad=True)

        def map_fb():
            clip.zero_grad(set_to_none=True)
            xg.grad = None
            clip(xg).square().mean().backward()

        ia = torch.randn(1024, 384, device=device, requires_grad=True)
        ib = torch.randn(1024, 384, device=device, requires_grad=True)
        ls = torch.tensor(2.659, device=device, requires_grad=True)

        def nce():
            ia.grad = ib.grad = ls.grad = None
            clip_auxiliary_loss(ia, ib, ls).backward()

        out["cla_head_B1024"] = {"ms_map_block_fwd_bwd": round(timed(map_fb, 5, 30) * 1e3, 3),
                                 "ms_infonce_value_and_grads": round(timed(nce, 5, 30) * 1e3, 3),
                                 "what": "ClipStyleProjection('map') forward + backward on (1024, 4, 384) context tokens, and the InfoNCE op "
                                         "(value + gradients of both embeddings and the logit scale) at B = 1024"}
    except Exception as e:  # diagnostic leg
        out["cla_head_B1024"] = {"error": repr(e)}
    return out


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, P, batch, n_denoise, budget_s, threads, only=None):
    """Oracle (port of the reference's PyTorch-CPU algorithm) on this host.  Legs (BASELINE.md section 3 / SURVEY.md
    8(d)): all host threads, the measured-best thread count and ONE thread; as written (encoder re-run every step, like
    the reference) and with the encoder hoisted.  `value` / `cores` quote the as-written leg at the best thread count.
    Bounded: every leg probes at B=16 and sizes its sample so that the whole baseline stays inside ``budget_s``."""
    from mdt_policy_amd import synthetic
    from oracle import mdt_oracle as O
    inp = {k: torch.from_numpy(v) for k, v in synthetic.sampler_inputs(batch, cfg, seed=1).items()}
    sig = O.get_sigmas_exponential(n_denoise, 0.001, 80.0)
    x = inp["noise"] * 80.0
    ncpu = os.cpu_count() or 1

    def run(b, hoist):
        st = {"state_images": inp["state_images"][:b], "modality": "lang"}
        t0 = time.perf_counter()
        O.sample_ddim(P, cfg, st, x[:b], inp["goal"][:b], sig, hoist=hoist)
        return time.perf_counter() - t0

    def leg(nthreads, hoist, budget):
        """One timed configuration, strictly inside `budget` seconds: a B=16 run doubles as warm-up and probe; a leg whose
        probe already eats the budget (all 256 host threads on these tiny matrices) reports the probe itself."""
        torch.set_num_threads(nthreads)
        t_leg = time.perf_counter()
        first = run(16, hoist)  # warm-up (thread pool, allocator)
        b, best, reps = 16, first, 1
        if first < budget / 6:
            probe = run(16, hoist)
            best = min(best, probe)
            b = batch
            while b > 16 and probe * (b / 16) * 1.5 > budget - (time.perf_counter() - t_leg):
                b //= 2
            if b > 16:
                best, reps = float("inf"), 0
                while reps < 3 and (reps == 0 or (time.perf_counter() - t_leg) + best < budget):
                    best, reps = min(best, run(b, hoist)), reps + 1
        return {"threads": nthreads, "mode": "hoisted" if hoist else "as_written", "value": round(b / best, 2), "sample_batch": b,
                "seconds": round(best, 3), "reps": reps}

    if only is not None:  # child process: exactly one leg
        if len(only) > 3 and only[3]:  # worker of the all-host-threads leg: warm up, report, wait for "go", one timed run
            if len(only) > 4 and only[4] >= 0 and hasattr(os, "sched_setaffinity"):
                # its own block of logical CPUs (what numactl / taskset do for a batch-sharded deployment): unpinned, 16 x 16
                # OpenMP threads migrate over 256 CPUs and spin in each other's barriers (measured 88 chunks/s for the host)
                cpus = sorted(os.sched_getaffinity(0))
                mine = cpus[only[4] * only[0]:(only[4] + 1) * only[0]]
                if len(mine) == only[0]:
                    os.sched_setaffinity(0, mine)
            torch.set_num_threads(only[0])
            run(16, bool(only[1]))
            print("READY", flush=True)
            sys.stdin.readline()
            return {"seconds": run(int(only[3]), bool(only[1])), "sample_batch": int(only[3])}
        return leg(only[0], bool(only[1]), float(only[2]))
    import subprocess
    plan = [(threads, False), (1, False), (threads, True), (1, True)]
    legs, seen = [], set()
    for nt, hoist in plan:
        if (nt, hoist) in seen:
            continue
        seen.add((nt, hoist))
        per = budget_s / (len(plan) + 1)
        # every leg runs in its own process under a hard time limit: with all host threads the oracle's tiny matrices
        # spend their time in thread barriers and a single call can take minutes -- such a leg is reported as unfinished
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", f"{nt},{int(hoist)},{per}", "--batch", str(batch),
               "--denoise-steps", str(n_denoise)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=2.0 * per + 20, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            legs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
        except (subprocess.TimeoutExpired, IndexError, ValueError) as e:
            legs.append({"threads": nt, "mode": "hoisted" if hoist else "as_written", "value": None,
                         "note": f"did not finish within {2.0 * per + 20:.0f} s ({type(e).__name__})"})
        log(f"cpu baseline leg {legs[-1]}")
    # ALL host threads.  One process with every thread spends its time in thread barriers on these small
    # matrices (a B = 16 probe at 256 threads does not finish in 30 s); the way this workload uses a whole host is the way it
    # uses N GPUs: shard the batch.  P worker processes of `threads` threads each (P * threads = host threads) sample their
    # own shard at the same time; value = all chunks / the wall time from the common start to the last worker's finish.
    nproc = ncpu // max(1, threads)
    if nproc >= 2 and legs[0].get("value"):
        shard = max(16, min(batch, int(legs[0]["value"] * 2) // 16 * 16))  # ~2 s per worker at the single-process rate
        cmd = lambda i: [sys.executable, os.path.abspath(__file__), "--cpu-leg", f"{threads},0,0,{shard},{i}", "--batch", str(batch),
                         "--denoise-steps", str(n_denoise)]
        procs = [subprocess.Popen(cmd(i), stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                  env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_WAIT_POLICY="PASSIVE", OMP_PROC_BIND="false"))
                 for i in range(nproc)]
        entry = {"threads": ncpu, "mode": "as_written", "processes": nproc, "threads_per_process": threads, "sample_batch": shard * nproc,
                 "pinning": "each worker on its own block of logical CPUs (sched_setaffinity)"}
        try:
            import select
            deadline = time.perf_counter() + 120.0
            for pr in procs:  # every worker has imported torch and run its warm-up
                while True:
                    if not select.select([pr.stdout], [], [], max(0.0, deadline - time.perf_counter()))[0]:
                        raise TimeoutError("workers not ready within 120 s")
                    line = pr.stdout.readline()
                    if line.startswith("READY") or not line:
                        break
            t0 = time.perf_counter()
            for pr in procs:
                pr.stdin.write("go\n"); pr.stdin.flush()
            outs = []
            for pr in procs:
                if not select.select([pr.stdout], [], [], max(0.0, t0 + 120.0 - time.perf_counter()))[0]:
                    raise TimeoutError("a worker did not finish within 120 s")
                outs.append(json.loads(pr.stdout.readline()))
            wall = time.perf_counter() - t0
            entry.update(value=round(shard * nproc / wall, 2), seconds=round(wall, 3), reps=1,
                         slowest_worker_s=round(max(o["seconds"] for o in outs), 3))
        except Exception as e:
            entry.update(value=None, note=f"{type(e).__name__}: {e}")
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        legs.insert(1, entry)
        log(f"cpu baseline leg {entry}")
    else:
        legs.insert(1, {"threads": ncpu, "mode": "as_written", "value": legs[0].get("value"),
                        "note": f"host has {ncpu} threads: the {threads}-thread leg IS the all-threads leg"})
    head = legs[0]
    return {"value": head["value"], "unit": "action-chunks/s", "cores": head["threads"], "kind": "port",
            "sample": f"oracle sample_ddim, B={head['sample_batch']}, {n_denoise} steps, fp32, encoder re-run every step "
                      f"(as the reference does), best of {head['reps']} ({head['seconds']:.2f} s each), {head['threads']} of "
                      f"{ncpu} host threads, torch {torch.__version__}",
            "cpu_model": cpu_model(), "host_threads": ncpu, "legs": legs}


def main():
    args = parse()
    if args.cpu_leg:  # child of cpu_baseline(): CPU only, no GPU, no distributed
        from mdt_policy_amd import configs, synthetic
        nt, hoist, budget, *rest = args.cpu_leg.split(",")
        shard = int(rest[0]) if rest else 0
        widx = int(rest[1]) if len(rest) > 1 else -1
        cfg = configs.mdtv_default()
        from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
        shapes = [(k, tuple(v.shape)) for k, v in GCDenoiser(cfg, sigma_data=0.5).state_dict().items()]
        P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=0, profile="init").items()}
        print(json.dumps(cpu_baseline(cfg, P, args.batch, args.denoise_steps, float(budget), int(nt), only=(int(nt), int(hoist), float(budget), shard, widx))), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU: the hot path has no CPU execution path")
    # MDT_BENCH_SHARE_GPU=1 (+ MDT_BENCH_BACKEND=gloo) lets the N > 1 code path be exercised on a 1-GPU box: all ranks
    # use device 0 and gather through gloo.  Never set by the driver; numbers from such a run are meaningless.
    share = os.environ.get("MDT_BENCH_SHARE_GPU") == "1"
    device = torch.device("cuda", 0 if share else local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MDT_BENCH_BACKEND", "nccl")  # "nccl" is RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mdt_policy_amd import sharding, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs

    cfg, P, model = build_model(device)
    B = args.batch
    inp = {k: torch.from_numpy(v).to(device) for k, v in synthetic.sampler_inputs(B, cfg, seed=1 + rank).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    goal, x_T = inp["goal"], inp["noise"] * 80.0
    sigmas = gs.get_sigmas_exponential(args.denoise_steps, 0.001, 80.0)  # host tensor, like the reference's cpu default
    eng = model.inner_model.hip_engine(0.5)
    eng.reserve(B)

    gather_evs = []  # N > 1: a pair of HIP events around the collective of every TIMED step (recorded on the launch stream:
    #                  what they bracket is the gather as the stream sees it, waiting for the slowest peer included)
    timing_gather = [False]

    def step():
        with torch.no_grad():
            act = gs.sample_ddim(model, state, x_T, goal, sigmas)
        if world > 1:
            if timing_gather[0]:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
            act = sharding.all_gather_actions(act, B * world)  # ONE collective per sample call
            if timing_gather[0]:
                g1.record()
                gather_evs.append((g0, g1))
        return act

    log(f"model ready on {device}; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timing_gather[0] = True
    # shader-clock / 100 MHz counter pairs of one wave in front of and behind the timed region (mdt_op_clock_stamp): the clock the
    # chip SUSTAINED over exactly these steps -- `frac` is quoted against the peak at the 2.4 GHz specification clock
    from mdt_policy_amd import _lib as _mdt_lib
    stamps = torch.zeros(2, 16, dtype=torch.int64, device=device)
    lstream = torch.cuda.current_stream(device).cuda_stream
    t0 = time.perf_counter()
    e0.record()
    _mdt_lib.check(_mdt_lib.load().mdt_op_clock_stamp(stamps[0].data_ptr(), lstream))
    for _ in range(args.steps):
        out = step()
    _mdt_lib.check(_mdt_lib.load().mdt_op_clock_stamp(stamps[1].data_ptr(), lstream))
    e1.record()
    torch.cuda.synchronize(device)
    own_wall = time.perf_counter() - t0  # this rank's own clock, before it waits for the others
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    timing_gather[0] = False
    gpu_s = e0.elapsed_time(e1) * 1e-3
    per_rank = None
    if dist is not None:
        # what a shortfall at N > 1 is made of: every rank's own step time, its GPU time, and the collective as its stream saw it
        g_us = sorted(a.elapsed_time(b) * 1e3 for a, b in gather_evs)
        mine = torch.tensor([own_wall / args.steps * 1e3, gpu_s / args.steps * 1e3, g_us[len(g_us) // 2], g_us[-1]],
                            device=device, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu()
        t = torch.tensor([wall, gpu_s], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, gpu_s = t[0].item(), t[1].item()
    assert torch.isfinite(out).all()
    log(f"timed {args.steps} steps: wall {wall:.4f}s gpu {gpu_s:.4f}s")
    # spread (median of >= 50 timed iterations): the same step again, every call between its own pair of
    # HIP events on the launch stream; `value` stays the contract's K-step mean, this tells box-to-box noise from change
    n_spread = max(50, args.steps)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_spread)]
    for a, b in evs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize(device)
    per_call = torch.tensor(sorted(a.elapsed_time(b) for a, b in evs), dtype=torch.float64)
    if dist is not None:
        dist.barrier()
        pc = per_call.to(device)
        dist.all_reduce(pc, op=dist.ReduceOp.MAX)  # rank-wise order statistics: the slowest rank at every quantile
        per_call = pc.cpu()
    q = lambda f: float(per_call[min(n_spread - 1, int(round(f * (n_spread - 1))))])
    spread = {"calls": n_spread, "median_ms": round(q(0.5), 4), "p10_ms": round(q(0.1), 4), "p90_ms": round(q(0.9), 4),
              "min_ms": round(float(per_call[0]), 4), "max_ms": round(float(per_call[-1]), 4), "mean_ms": round(float(per_call.mean()), 4),
              "what": "per-call HIP-event times of one more run of the same step (each call bracketed on the launch stream)"}
    gather_ok = None
    if dist is not None and os.environ.get("MDT_BENCH_VERIFY_GATHER") == "1":
        # every rank checks the gathered tensor: shape (world * B, Ta, A) and block r == rank r's own actions
        with torch.no_grad():
            mine = gs.sample_ddim(model, state, x_T, goal, sigmas)
        blocks = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(blocks, mine)
        ok = out.shape == (B * world,) + tuple(mine.shape[1:]) and all(
            torch.equal(out[r * B:(r + 1) * B], blocks[r]) for r in range(world))
        flag = torch.tensor([1 if ok else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_ok = bool(flag.item())

    if rank == 0:
        flops_chunk = eng.flops_per_chunk(args.denoise_steps)
        chunks = B * world * args.steps
        achieved = flops_chunk * B * args.steps / gpu_s / 1e12  # per GPU, HIP-event time of the timed region
        res = {
            "metric": "action-chunks/sec (10-step denoise, horizon=10)",
            "value": round(chunks / wall, 1),
            "unit": "action-chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "median_ms": spread["median_ms"], "p10_ms": spread["p10_ms"], "p90_ms": spread["p90_ms"],
            "value_at_median": round(B * world / (spread["median_ms"] * 1e-3), 1),
            "spread": spread,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",   # fp32 in, fp32 out, fp32 accumulation; the MLP / qkv launches multiply three-way bf16 splits of the fp32 operands (fp32 product accuracy)
            "data": "synthetic",
            "config": {"workload": f"MDT-V d=384 4+4 blocks 8 heads, horizon 10, action dim 7, "
                                   f"{args.denoise_steps} DDIM steps (exponential sigma 80->0.001), "
                                   f"B={B} synthetic goal/state tokens per GPU (BASELINE configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "denoise_steps": args.denoise_steps,
                       "parallelism": f"batch-sharded x{world}, RCCL all-gather of actions" if world > 1 else "single GPU",
                       "weights": "random init N(0,0.02) (reference _init_weights distributions), seed 0"},
            "collective": ({"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                            "ranks": dist.get_world_size(), "per_step": "one all_gather_into_tensor of (B, 10, 7) fp32",
                            "gather_verified": gather_ok,
                            "bytes_per_rank": int(B * out.shape[1] * out.shape[2] * 4),
                            # median over the timed steps of the HIP-event time around all_gather_actions, per rank; the figure of
                            # the slowest rank is the one quoted (a fast rank's gather also contains its wait for the slowest)
                            "gather_us": round(float(per_rank[:, 2].min()), 1),
                            "gather_us_per_rank": [round(float(v), 1) for v in per_rank[:, 2]],
                            "gather_us_max_per_rank": [round(float(v), 1) for v in per_rank[:, 3]],
                            "gather_what": "HIP events on the launch stream around the collective of every timed step: median per rank; "
                                           "`gather_us` = the smallest median = the rank that arrives last and waits for nobody",
                            "per_rank_ms": [round(float(v), 4) for v in per_rank[:, 0]],
                            "per_rank_gpu_ms": [round(float(v), 4) for v in per_rank[:, 1]],
                            "slowest_rank": int(per_rank[:, 0].argmax())}
                           if dist is not None else None),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                         "scope": "whole sampler call (encoder + 10 decoder steps), algorithmic "
                                  f"{flops_chunk / 1e9:.3f} GFLOP/chunk, HIP-event time of the timed region",
                         "peak_scope": "the dense FP32 matrix peak (v_mfma_f32_16x16x4_f32), the rate of the path's contract: fp32 operands, "
                                       "fp32 results.  Since round 6 the two dominant launches (fused MLP, qkv) multiply three-way bf16 splits "
                                       "of their fp32 operands (six bf16 MFMA products per k32 step, fp32 accumulation: fp32 product accuracy "
                                       "at 2.7x the fp32 matrix rate), so `frac` is no longer bounded by 1; dominant_kernel."
                                       "frac_of_bf16_peak_executed prices that launch's executed bf16 MFMA FLOPs against the 2.5 PFLOP/s bf16 peak",
                         "gpu_ms_per_step": round(gpu_s / args.steps * 1e3, 4)},
        }
        try:  # the sustained clock of the timed region and the fraction of the peak AT that clock
            mhz, span_us = sustained_mhz(stamps)
            if mhz:
                d_ref = span_us * 100.0
                res["roofline"]["sustained_mhz"] = round(mhz, 1)
                res["roofline"]["spec_mhz"] = SPEC_MHZ
                res["roofline"]["frac_at_sustained_clock"] = round(achieved / (PEAK_FP32_MFMA_TFLOPS * mhz / SPEC_MHZ), 4)
                res["roofline"]["clock_scope"] = ("s_memtime / s_memrealtime (100 MHz) of one wave per XCD stamped in front of and behind the timed "
                                                  f"region on the launch stream ({d_ref / 100.0:.0f} us apart): the average shader clock over "
                                                  "exactly the timed steps (median over the XCDs); `frac` stays quoted at the 2.4 GHz the 157.3 TFLOP/s peak assumes")
        except Exception as e:  # diagnostic only
            res["roofline"]["sustained_mhz_error"] = repr(e)
        try:
            dk = time_dominant_kernel(device, B * 10)
            chain = time_dominant_kernel_in_chain(step, device) if world == 1 else None
            if chain:  # the figure the roofline of the dominant kernel is quoted on: its launches inside one sampler call
                tf = dk["gflop_per_launch"] * 1e9 / (chain["avg_us"] * 1e-6) / 1e12
                if dk.get("executed_flops_per_algorithmic", 1) > 1:
                    dk["frac_of_bf16_peak_executed"] = round(dk["executed_flops_per_algorithmic"] * tf / PEAK_BF16_MFMA_TFLOPS, 4)
                dk.update({"avg_us": chain["avg_us"], "achieved": round(tf, 2), "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                           "in_chain": chain,
                           "what": f"mean over the {chain['launches']} launches of ONE sampler call, each between its own pair of HIP "
                                   "events on the launch stream (mdt_op_trace_mlp), minus the cost of an empty event bracket measured "
                                   f"in the same call ({chain['empty_bracket_us']} us; the raw brackets average {chain['avg_us_bracket']} us): "
                                   "the kernel as a kernel trace of the call sees it"})
            else:
                dk.update({k: dk["alone"][k] for k in ("avg_us", "achieved", "frac")})
            res["roofline"]["dominant_kernel"] = dk
            log("dominant kernel timed")
            # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process, so `traffic` is NOT
            # a measurement of this run: it is the committed rocprofv3 --pmc pass of the same kernel and shape (latest profiles/
            # round), labelled with the commit that pass was taken at
            here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            cands = sorted(f for f in os.listdir(here) if f.endswith("_dominant_kernel_pmc.json")) if os.path.isdir(here) else []
            if B == 256 and cands:
                with open(os.path.join(here, cands[-1])) as f:
                    j = json.load(f)
                res["roofline"]["traffic"] = j["hbm_side_bytes_per_launch"]
                # the EXECUTED share of the matrix pipe, beside the algorithmic fraction above (which credits the folded cross-
                # attention projections and the per-step conditioning at the reference's FLOP count): whole-run MFMA-busy of the
                # same committed PMC pass (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel time at 2.4 GHz)
                if "whole_run_mfma_busy" in j:
                    res["roofline"]["mfma_busy"] = j["whole_run_mfma_busy"]
                    res["roofline"]["mfma_busy_scope"] = (
                        f"NOT measured by this run: whole-run MFMA-busy fraction of the committed PMC pass profiles/{cands[-1]} "
                        "(executed MFMA issue slots / peak slots at 2.4 GHz over all kernels of the bench command); `frac` above counts "
                        "the reference's algorithmic FLOPs, of which the folded cross-attention projections and the per-sample sigma / adaLN "
                        "products are not executed")
                res["roofline"]["traffic_scope"] = (
                    f"NOT measured by this run: committed PMC pass profiles/{cands[-1]} (commit {j.get('commit', 'see its header')}), HBM-side "
                    "bytes per launch of the dominant kernel (TCC_EA0 read x128 B + write x64 B); true algorithmic bytes "
                    f"{dk['algorithmic_bytes_per_launch']} (rows in + both weight images + one output), "
                    f"{dk['algorithmic_bytes_per_launch_with_slabs']} with the three partial slabs the kernel writes instead of one output")
        except Exception as e:  # diagnostic leg only; never hides the main number
            res["roofline"]["dominant_kernel"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:  # diagnostic legs only; never hide the main number
                res["other_configs"] = other_configs(device)
                log("other configurations timed")
            except Exception as e:
                res["other_configs"] = {"error": str(e)}
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            res["cpu_baseline"] = cpu_baseline(cfg, P, B, args.denoise_steps, args.cpu_seconds, threads)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
