"""Training path: gradients of GCDenoiser.loss (+ a scalar hung on latent_encoder_emb).
CPU: autograd through the oracle vs the golden gradient summaries of the REFERENCE's own loss.backward()
(tests/golden/g11_grads_*.npz).  GPU: the HIP forward/backward behind torch.autograd vs both."""
import numpy as np
import pytest
import torch

from mdt_policy_amd import synthetic
from oracle import mdt_oracle as O
from tests.helpers import assert_close, cfg_of, inputs_of, load_fixture, params_of

CASES = ["mdtv_tiny", "mdt_tiny", "mdtv_bias_plain_goal", "mdtv_default", "mdtv_rope", "mdt_rope", "mdtv_noise_block",
         "mdtv_no_ada", "mdt_no_ada", "mdtv_mlp_head", "mdtv_no_goal_cond", "mdt_no_goal_cond"]


def case(name):
    meta, fx = load_fixture(f"g11_grads_{name}.npz")
    cfg = cfg_of(meta)
    state, goal, _ = inputs_of(meta)
    li = {k: torch.from_numpy(v) for k, v in synthetic.loss_inputs(meta["B"], cfg, meta["loss_seed"]).items()}
    return meta, fx, cfg, state, goal, li


def summary(g):
    g = g.detach().double().cpu()
    return [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]


def check_summaries(got, want, what, rtol=2e-3):
    """got / want: {name: [norm, sum, first 6]}; compared relative to the gradient's own scale."""
    assert set(k for k, v in want.items() if v is not None) == set(got), what
    for k, w in want.items():
        if w is None:
            continue
        g = np.array(got[k]); w = np.array(w)
        # floor: gradients that are mathematically zero (key biases: softmax is shift invariant) are fp32 noise
        tol = rtol * abs(w[0]) + 1e-6
        assert abs(g[0] - w[0]) <= tol, f"{what} {k}: norm {g[0]} vs {w[0]}"
        assert np.all(np.abs(g[1:] - w[1:]) <= tol), f"{what} {k}: {g[1:]} vs {w[1:]}"


def oracle_total(P, cfg, meta, state, goal, li, dtype):
    """loss + 0.1 * <ctx, w> / numel, as the fixture was generated."""
    st = {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in state.items()}
    loss, _ = O.loss(P, cfg, st, li["actions"].to(dtype), goal, li["noise_train"].to(dtype), li["sigma"].to(dtype),
                     arch=meta["arch"])
    entry = "forward"
    ctx = O.encode(P, cfg, st, goal, meta["arch"], entry, sigma=li["sigma"].to(dtype))
    wctx = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), meta["ctx_seed"])).to(dtype)
    return loss, loss + 0.1 * (ctx * wctx).sum() / ctx.numel()


@pytest.mark.parametrize("name", CASES)
def test_oracle_autograd_matches_the_reference_gradients(name):
    meta, fx, cfg, state, goal, li = case(name)
    P = {k: v.double().requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
    state = {k: (v.double().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    goal = goal.double().requires_grad_()
    loss, total = oracle_total(P, cfg, meta, state, goal, li, torch.float64)
    total.backward()
    assert abs(loss.item() - float(np.asarray(fx["loss"]).reshape(-1)[0])) <= 1e-4 * abs(float(np.asarray(fx["loss"]).reshape(-1)[0]))
    got = {k[len("inner_model."):]: summary(v.grad) for k, v in P.items() if v.grad is not None}
    want = {k[len("inner_model."):]: v for k, v in meta["grads"].items()}
    # the oracle holds lang_emb as an alias of goal_emb when there is no modality encoder: fold its gradient in
    check_summaries({k: v for k, v in got.items() if k in want and want[k] is not None}, want, name)
    for k, v in state.items():
        if torch.is_tensor(v):
            assert_close(v.grad, fx["d_" + k], rtol=2e-3, atol=1e-7, what="d_" + k)
    g_goal = goal.grad if goal.grad is not None else torch.zeros_like(goal)  # unused goal (MDT, goal_conditioned=False)
    assert_close(g_goal, fx["d_goal"], rtol=2e-3, atol=1e-7, what="d_goal")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_gradients_match_reference_and_oracle(name):
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case(name)
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()  # eval(): dropout off (the fixture's reference run was in eval mode too)
    gstate = {k: (v.cuda().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    ggoal = goal.cuda().requires_grad_()
    loss, mo = model.loss(gstate, li["actions"].cuda(), ggoal, li["noise_train"].cuda(), li["sigma"].cuda())
    ctx = model.inner_model.latent_encoder_emb
    wctx = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), meta["ctx_seed"])).cuda()
    total = loss + 0.1 * (ctx * wctx).sum() / ctx.numel()
    total.backward()
    assert abs(loss.item() - float(np.asarray(fx["loss"]).reshape(-1)[0])) <= 1e-3 * abs(float(np.asarray(fx["loss"]).reshape(-1)[0]))
    got = {k: summary(p.grad) for k, p in model.inner_model.named_parameters() if p.grad is not None}
    want = {k[len("inner_model."):]: v for k, v in meta["grads"].items()}
    check_summaries(got, want, name + " vs reference")
    for k, v in gstate.items():
        if torch.is_tensor(v):
            assert_close(v.grad.cpu(), fx["d_" + k], rtol=2e-3, atol=1e-6, what="d_" + k)
    g_goal = ggoal.grad.cpu() if ggoal.grad is not None else torch.zeros_like(goal)
    assert_close(g_goal, fx["d_goal"], rtol=2e-3, atol=1e-6, what="d_goal")
    # full tensors against float64 autograd through the oracle
    P = {k: v.double().requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
    st64 = {k: (v.double() if torch.is_tensor(v) else v) for k, v in state.items()}
    _, tot64 = oracle_total(P, cfg, meta, st64, goal.double(), li, torch.float64)
    tot64.backward()
    for k, p in model.inner_model.named_parameters():
        ref = P["inner_model." + k].grad
        if ref is None:
            assert p.grad is None, k
            continue
        scale = float(ref.abs().max())
        assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * scale + 1e-7, what=k)  # floor: key-bias gradients are exactly 0 in theory


@pytest.mark.gpu
def test_hip_gradients_at_the_c3_batch_b1024():
    """BASELINE configs[2] at its own size: ONE MDT-V default training step at B = 1024 (M = 10 240 action rows) takes
    code the B <= 8 fixtures never reach -- the batched split-K dW products (~1000 workgroups of >= 512-row slices), the
    64-slice column sums and the co-resident 4-wave tiles of M >= 4096.  Every parameter gradient and the input
    gradients against float64 autograd through the oracle on the same seeded batch (eval mode: dropout off)."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, _, _, _ = case("mdtv_default")
    B = 1024
    state, goal, _ = inputs_of(meta, batch=B)
    li = {k: torch.from_numpy(v) for k, v in synthetic.loss_inputs(B, cfg, meta["loss_seed"]).items()}
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    gstate = {k: (v.cuda().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    ggoal = goal.cuda().requires_grad_()
    loss, mo = model.loss(gstate, li["actions"].cuda(), ggoal, li["noise_train"].cuda(), li["sigma"].cuda())
    ctx = model.inner_model.latent_encoder_emb
    wctx = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), meta["ctx_seed"])).cuda()
    (loss + 0.1 * (ctx * wctx).sum() / ctx.numel()).backward()
    torch.cuda.synchronize()
    # float64 autograd through the oracle (the checker) -- on the GPU's fp64 units when the oracle's ops run there,
    # else on the host
    def oracle_grads(dev):
        P = {k: v.double().to(dev).requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
        st = {k: (v.double().to(dev).requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
        g64 = goal.double().to(dev).requires_grad_()
        l64 = {k: v.to(dev) for k, v in li.items()}
        lo, tot = oracle_total_dev(P, cfg, meta, st, g64, l64, dev)
        tot.backward()
        return P, st, g64, lo
    def oracle_total_dev(P, cfg, meta, st, g, l, dev):
        lo, _ = O.loss(P, cfg, st, l["actions"].double(), g, l["noise_train"].double(), l["sigma"].double(), arch=meta["arch"])
        c = O.encode(P, cfg, st, g, meta["arch"], "forward", sigma=l["sigma"].double())
        w = torch.from_numpy(synthetic.normal("ctx_weight", tuple(c.shape), meta["ctx_seed"])).double().to(dev)
        return lo, lo + 0.1 * (c * w).sum() / c.numel()
    try:
        P, st64, g64, lo = oracle_grads("cuda")
    except Exception:  # an oracle op without a device kernel: run the checker on the host
        torch.set_num_threads(min(32, torch.get_num_threads() * 4))
        P, st64, g64, lo = oracle_grads("cpu")
    assert abs(loss.item() - lo.item()) <= 1e-3 * abs(lo.item())
    n = 0
    for k, p in model.inner_model.named_parameters():
        ref = P["inner_model." + k].grad
        if ref is None:
            assert p.grad is None, k
            continue
        scale = float(ref.abs().max())
        assert_close(p.grad.cpu(), ref.cpu(), rtol=2e-3, atol=2e-3 * scale + 1e-7, what=f"B=1024 {k}")
        n += 1
    assert n > 100
    for k, v in gstate.items():
        if torch.is_tensor(v):
            ref = st64[k].grad.cpu()
            assert_close(v.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-9, what=f"B=1024 d_{k}")
    assert_close(ggoal.grad.cpu(), g64.grad.cpu(), rtol=2e-3, atol=2e-3 * float(g64.grad.abs().max()) + 1e-9, what="B=1024 d_goal")


@pytest.mark.gpu
def test_hip_training_steps_follow_the_oracle():
    """Five AdamW steps on the facade (HIP forward/backward, torch optimizer) against the same five steps on the
    oracle with torch autograd: losses and a few weights stay together."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case("mdtv_tiny")
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
    used = [k for k in P if "proprio_emb" not in k and "rotary" not in k and k != "inner_model.pos_emb"]
    opt_h = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
    opt_o = torch.optim.AdamW([P[k] for k in used], lr=1e-3, weight_decay=0.05)
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    for step in range(5):
        opt_h.zero_grad(); opt_o.zero_grad()
        lh, _ = model.loss(gstate, li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())
        lh.backward(); opt_h.step()
        lo, _ = O.loss(P, cfg, state, li["actions"], goal, li["noise_train"], li["sigma"], arch="mdtv")
        lo.backward(); opt_o.step()
        assert abs(lh.item() - lo.item()) <= 2e-3 * abs(lo.item()), (step, lh.item(), lo.item())
    sd = model.state_dict()
    for k in ("inner_model.decoder.blocks.1.mlp.c_fc.weight", "inner_model.encoder.blocks.0.attn.query.weight",
              "inner_model.sigma_emb.1.weight", "inner_model.action_pred.weight"):
        assert_close(sd[k].cpu(), P[k].detach(), rtol=1e-3, atol=1e-4, what=k)
    assert lh.item() < float(np.asarray(fx["loss"]).reshape(-1)[0])  # it learns


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mdtv_tiny", "mdtv_no_ada", "mdtv_noise_block"])
def test_hip_context_only_backward_and_tape_rules(name):
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case(name)
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    gstate = {k: (v.cuda().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    # two forwards alive at once (the reference's training step: loss + forward_context_only, then ONE backward)
    loss, _ = model.loss(gstate, li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())
    ctx = model.forward_context_only(gstate, li["actions"].cuda(), goal.cuda(), li["sigma"].cuda())
    w = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), 7)).cuda()
    (loss + (ctx * w).mean()).backward()
    P = {k: v.double().requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
    st64 = {k: (v.double().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    l64, _ = O.loss(P, cfg, st64, li["actions"].double(), goal.double(), li["noise_train"].double(), li["sigma"].double())
    c64 = O.forward_context_only(P, cfg, st64, goal.double(), sigma=li["sigma"].double())
    (l64 + (c64 * w.cpu().double()).mean()).backward()
    for k in ("tok_emb.weight", "encoder.blocks.0.mlp.c_proj.weight", "lang_emb.2.bias", "decoder.blocks.0.cross_att.key.weight",
              "sigma_emb.1.weight", "sigma_emb.3.bias"):
        ref = P["inner_model." + k].grad
        got = dict(model.inner_model.named_parameters())[k].grad
        assert_close(got.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()), what=k)
    assert_close(gstate["state_images"].grad.cpu(), st64["state_images"].grad, rtol=2e-3, atol=1e-6, what="d_tokens")
    # under no_grad nothing is taped and the plain forward path answers
    with torch.no_grad():
        l2, _ = model.loss(gstate, li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())
    assert abs(l2.item() - loss.item()) < 1e-4 * abs(loss.item())


@pytest.mark.gpu
def test_hip_train_mode_dropout_is_seeded_consistent_and_differentiated_correctly():
    """train() mode with the shipped dropout rates (0.3 / 0.1 / 0.05).  torch's RNG stream cannot be matched, so:
    same torch seed -> identical loss and gradients; another seed -> different; and, with the masks pinned by the
    seed, the backward is the derivative of the forward (central finite differences along random directions)."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case("mdtv_tiny")
    assert cfg["attn_pdrop"] == 0.3 and cfg["resid_pdrop"] == 0.1 and cfg["mlp_pdrop"] == 0.05
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().train()
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    args = (li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())

    def run(seed, backward=True):
        torch.manual_seed(seed)
        model.zero_grad()
        loss, _ = model.loss(gstate, *args)
        if backward:
            loss.backward()
            return loss.item(), {k: p.grad.clone() for k, p in model.inner_model.named_parameters() if p.grad is not None}
        return loss.item(), None

    l1, g1 = run(11)
    l1b, g1b = run(11)
    l2, g2 = run(12)
    assert l1 == l1b and all(torch.equal(g1[k], g1b[k]) for k in g1)
    assert l1 != l2 and not torch.equal(g1["decoder.blocks.0.mlp.c_fc.weight"], g2["decoder.blocks.0.mlp.c_fc.weight"])
    model.eval()
    le, _ = model.loss(gstate, *args)
    model.train()
    assert abs(l1 - le.item()) > 1e-4 * abs(le.item())           # dropout really changes the forward
    # directional derivative with pinned masks; float32 forward: eps large enough to beat rounding
    params = dict(model.inner_model.named_parameters())
    gen = torch.Generator().manual_seed(5)
    for names in (["decoder.blocks.1.mlp.c_fc.weight", "decoder.blocks.0.attn.query.weight", "sigma_emb.3.weight"],
                  ["encoder.blocks.0.attn.value.weight", "tok_emb.weight", "decoder.blocks.1.cross_att.key.weight",
                   "decoder.blocks.0.adaLN_zero.modulation.1.weight", "decoder.ln.weight"]):
        dirs = {k: torch.randn(params[k].shape, generator=gen).cuda() for k in names}
        analytic = sum(float((g1[k] * dirs[k]).sum()) for k in names)
        eps = 2e-3
        with torch.no_grad():
            for k in names: params[k].add_(eps * dirs[k])
        lp, _ = run(11, backward=False)
        with torch.no_grad():
            for k in names: params[k].sub_(2 * eps * dirs[k])
        lm, _ = run(11, backward=False)
        with torch.no_grad():
            for k in names: params[k].add_(eps * dirs[k])
        numeric = (lp - lm) / (2 * eps)
        assert abs(numeric - analytic) <= 0.03 * abs(analytic) + 2e-4, (names, numeric, analytic)
    # tapes of forwards that never ran a backward were handed back (60+ forwards above, 16 tapes at most)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="dropout"):
        model.loss(gstate, *args)


def _ddp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the box's one GPU: RCCL cannot
    try:
        from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
        meta, fx, cfg, state, goal, li = case("mdtv_tiny")
        model = GCDenoiser(cfg, 0.5)
        model.load_state_dict(params_of(meta))
        model = model.cuda().eval()
        B = meta["B"]
        lo, hi = rank * B // world, (rank + 1) * B // world
        # find_unused_parameters: what Lightning 1.x's strategy "ddp" (reference mdt/training.py:74-79) passes, and what the reference
        # model needs as well -- pos_emb, proprio_emb and the other modality's goal embedder take no part in a forward
        # (their gradient stays None, and AdamW skips them instead of decaying them)
        ddp = DDP(_LossModule(model), find_unused_parameters=True, bucket_cap_mb=0.05)
        eng_of = lambda: next(iter(model.inner_model._engines.values()))
        log = []  # (bucket index, stages of the HIP backward enqueued when the bucket's reduction was launched)

        def hook(state_, bucket):
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            log.append((bucket.index(), eng_of().stages_enqueued))
            return default_hooks.allreduce_hook(None, bucket)

        ddp.register_comm_hook(None, hook)
        st = {k: (v[lo:hi].cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
        per_iter = []
        for _ in range(2):  # a second iteration: the reducer must have seen every bucket of the first one complete
            ddp.zero_grad()
            loss = ddp(st, li["actions"][lo:hi].cuda(), goal[lo:hi].cuda(), li["noise_train"][lo:hi].cuda(),
                       li["sigma"][lo:hi].cuda())
            s0 = eng_of().stages_enqueued
            del log[:]
            loss.backward()
            per_iter.append([(b, n - s0) for b, n in log])
        g = {k: p.grad.cpu().numpy() for k, p in model.inner_model.named_parameters() if p.grad is not None}
        q.put((rank, (g, per_iter, eng_of()._n_stages) if rank == 0 else None))
    finally:
        dist.destroy_process_group()


class _LossModule(torch.nn.Module):
    """What a LightningModule's training_step does with the denoiser: returns the diffusion loss."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, state, action, goal, noise, sigma):
        return self.model.loss(state, action, goal, noise, sigma)[0]


@pytest.mark.gpu
def test_hip_gradients_under_distributed_data_parallel():
    """DistributedDataParallel around the facade (2 processes, each half of the batch): the all-reduced gradients
    equal the single-process full-batch gradients.  gloo here because both ranks sit on the box's single GPU; on a
    node with one GPU per rank the same code runs over backend='nccl' (RCCL)."""
    import torch.multiprocessing as mp
    from tests.test_sharding_gloo import _free_port
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    meta, fx, cfg, state, goal, li = case("mdtv_tiny")
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    st = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    loss, _ = model.loss(st, li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())
    loss.backward()
    full = {k: p.grad.cpu() for k, p in model.inner_model.named_parameters() if p.grad is not None}
    grads, per_iter, n_stages = got[0]
    assert set(full) == set(grads)
    for k, ref in full.items():
        assert_close(grads[k], ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()) + 1e-7, what=k)
    # The backward ran in stages (a process group with two ranks exists: score_wrappers._staged_backward), one autograd node per
    # stage, so DDP launched bucket reductions while later stages had not been enqueued yet: at least six of them before the last
    # stage, in every iteration -- with the one-node backward of rounds 1-5 every hook saw the whole backward enqueued.
    for it in per_iter:
        early = [b for b, n in it if n < n_stages]
        assert len(it) >= 8 and len(early) >= 6, (n_stages, it)
        # Bucket 0 holds the LAST registered parameters (action_pred, action_emb, sigma_emb, decoder.ln, the last decoder
        # blocks) and DDP launches buckets in order: sigma_emb / action_emb need the whole decoder's backward, so the first
        # reduction leaves with the decoder's last stage (Ld + 1 stages enqueued) -- the encoder's stages and the embeddings
        # (a quarter of the backward) run under the reductions
        Ld = cfg["n_dec_layers"]
        assert min(n for _, n in it) <= Ld + 1, it


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mdtv_tiny", "mdtv_default", "mdt_tiny", "mdtv_noise_block", "mdtv_no_ada", "mdt_no_ada", "mdtv_mlp_head",
                                  "mdtv_bias_plain_goal", "mdt_no_goal_cond"])
def test_staged_backward_equals_the_one_call_backward_and_completes_parameters_stage_by_stage(name, monkeypatch):
    """mdt_train_loss_bwd_stage (round 6): (i) the chain of per-stage autograd nodes leaves the gradients of the one-node
    backward, bit for bit, including the input gradients and a gradient arriving at latent_encoder_emb; (ii) after stage k the
    gradient slots of every parameter with mdt_train_param_stage == k hold their final value (what a DDP bucket reads)."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case(name)
    res = {}
    for staged in ("0", "1"):
        monkeypatch.setenv("MDT_HIP_BWD_STAGES", staged)
        model = GCDenoiser(cfg, 0.5)
        model.load_state_dict(params_of(meta))
        model = model.cuda().eval()
        st = {k: (v.cuda().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
        g = goal.cuda().requires_grad_()
        loss, _ = model.loss(st, li["actions"].cuda(), g, li["noise_train"].cuda(), li["sigma"].cuda())
        extra = (model.inner_model.latent_encoder_emb * 0.37).sum()   # another loss hanging on the context
        (loss + extra).backward()
        res[staged] = ({k: p.grad.clone() for k, p in model.inner_model.named_parameters() if p.grad is not None},
                       {k: v.grad.clone() for k, v in st.items() if torch.is_tensor(v) and v.grad is not None}, g.grad.clone())
    (p0, i0, g0), (p1, i1, g1) = res["0"], res["1"]
    assert set(p0) == set(p1) and set(i0) == set(i1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), f"{k}: staged and one-call gradients differ"
    for k in i0:
        assert torch.equal(i0[k], i1[k]), f"input gradient {k} differs"
    assert torch.equal(g0, g1)
    # (ii) stage by stage through the engine
    st = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    eng = model._engine(allow_grad=True, state=st)
    tok, tok2, gg, B, _, _ = model._train_inputs(eng, st, goal.cuda(), model.inner_model._arch == "mdtv")
    a, nz = eng._in(li["actions"].cuda(), (B, eng.Ta, eng.A)), eng._in(li["noise_train"].cuda(), (B, eng.Ta, eng.A))
    _, _, _, tape = eng.train_loss_fwd(st, tok, tok2, gg, a, nz, eng._in(li["sigma"].cuda(), (B,)), None)
    buf = eng.train_loss_bwd_begin(torch.ones((), device="cuda"), None, tok, tok2, gg, (False, False, False))
    snaps = []
    for k in range(eng._n_stages):
        eng.train_loss_bwd_stage(tape, k, *buf)
        torch.cuda.synchronize()
        snaps.append(buf[0].clone())
    eng.tape_release(tape)
    final = snaps[-1]
    stages_seen = set()
    for nm, (off, n) in eng._grad_layout.items():
        k = eng._param_stage[nm]
        stages_seen.add(k)
        assert torch.equal(snaps[k][off:off + n], final[off:off + n]), f"{nm}: not complete behind its stage {k}"
    assert len(stages_seen) >= eng._n_stages - 1   # every stage (but possibly one) completes some parameter


def _dropout_stats(overrides, fixture="g13_dropout_stats.npz"):
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx = load_fixture(fixture)
    cfg = dict(cfg_of(meta), **overrides)
    state, goal, _ = inputs_of(meta)
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(meta["B"], cfg, meta["loss_seed"]).items()}
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().train()
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    n = meta["n"]
    losses, outs = [], []
    for i in range(n):
        torch.manual_seed(5000 + i)
        loss, mo = model.loss(gstate, li["actions"], goal.cuda(), li["noise_train"], li["sigma"])
        losses.append(loss.item()); outs.append(mo.detach())
        del loss
    outs = torch.stack(outs).double().cpu()
    return meta, fx, n, float(np.mean(losses)), float(np.std(losses, ddof=1)), outs.mean(0).numpy(), outs.var(0, unbiased=True).numpy()


def _check_dropout_stats(meta, fx, n, lm, ls, mean, var):
    se = (meta["loss_std"] ** 2 / n + ls ** 2 / n) ** 0.5
    assert abs(lm - meta["loss_mean"]) <= 4 * se, (lm, meta["loss_mean"], se)
    assert 0.8 <= ls / meta["loss_std"] <= 1.25, (ls, meta["loss_std"])
    assert abs(float(var.mean()) / meta["out_var_mean"] - 1) < 0.1, (float(var.mean()), meta["out_var_mean"])
    # element-wise: means within 5 standard errors (+ a small floor), variances within what the chi-square allows
    se_el = np.sqrt((fx["out_var"] + var) / n)
    assert np.all(np.abs(mean - fx["out_mean"]) <= 5 * se_el + 1e-3)
    ratio = (var + 1e-6) / (fx["out_var"] + 1e-6)
    assert np.all((ratio > 0.6) & (ratio < 1.6)), (ratio.min(), ratio.max())


@pytest.mark.gpu
def test_hip_dropout_statistics_match_the_reference():
    """The dropout streams of two implementations cannot coincide, their distributions must: 400 seeded train-mode
    forward passes on fixed weights and inputs against the same statistics of the REFERENCE (g13): mean / spread of the
    loss, and mean / variance over seeds of the model output."""
    meta, fx, n, lm, ls, mean, var = _dropout_stats({})
    _check_dropout_stats(meta, fx, n, lm, ls, mean, var)
    assert abs(lm - meta["loss_eval"]) > 2 * (meta["loss_std"] ** 2 / n + ls ** 2 / n) ** 0.5   # not the eval-mode loss


@pytest.mark.gpu
@pytest.mark.parametrize("overrides", [dict(attn_pdrop=0.0), dict(resid_pdrop=0.0), dict(mlp_pdrop=0.6)])
def test_hip_dropout_statistics_detect_a_wrong_site(overrides):
    """Negative control: the same check rejects a model whose dropout differs at one site."""
    with pytest.raises(AssertionError):
        _check_dropout_stats(*_dropout_stats(overrides))


@pytest.mark.gpu
def test_hip_embedding_dropout_and_goal_masking_statistics_match_the_reference():
    """embed_pdrob (self.drop on MDT's embedded goal / state / action tokens) and goal_drop (mask_cond) against 400
    train-mode passes of the REFERENCE with only those two switched on (g13_embed_goal_drop_stats)."""
    fixture = "g13_embed_goal_drop_stats.npz"
    meta, fx, n, lm, ls, mean, var = _dropout_stats({}, fixture)
    _check_dropout_stats(meta, fx, n, lm, ls, mean, var)
    # controls chosen for decisive margins (tools/dropout_margin.py prints every criterion against its threshold)
    for wrong in (dict(embed_pdrob=0.0), dict(goal_drop=0.6), dict(embed_pdrob=0.4)):
        with pytest.raises(AssertionError):
            _check_dropout_stats(*_dropout_stats(wrong, fixture))


@pytest.mark.gpu
def test_hip_embedding_dropout_is_differentiated_correctly():
    """Same seed -> same masks in forward and backward: the directional derivative of the seeded train-mode loss
    (embed_pdrob only, MDT: context and action embeddings) matches central finite differences."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case("mdt_tiny")
    cfg = dict(cfg, embed_pdrob=0.25, attn_pdrop=0.0, resid_pdrop=0.0, mlp_pdrop=0.0)
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().train()
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    args = (li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())

    def run(seed):
        torch.manual_seed(seed)
        return model.loss(gstate, *args)[0]

    l1, l2, l3 = run(3), run(3), run(4)
    assert l1.item() == l2.item() and l1.item() != l3.item()
    l1.backward()
    names = ["tok_emb.weight", "incam_embed.weight", "goal_emb.2.weight", "action_emb.weight", "encoder.blocks.0.attn.query.weight"]
    params = dict(model.inner_model.named_parameters())
    for k in names:
        p = params[k]
        g = p.grad.clone()
        d = torch.from_numpy(synthetic.normal("dir_" + k, tuple(p.shape), 9)).cuda()
        eps = 2e-2 / float(d.norm()) * max(float(p.detach().norm()), 1.0)
        with torch.no_grad():
            p.add_(eps * d)
        lp = run(3).item()          # train-mode dropout only runs under autograd
        with torch.no_grad():
            p.sub_(2 * eps * d)
        lm = run(3).item()
        with torch.no_grad():
            p.add_(eps * d)
        fd = (lp - lm) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) <= 3e-2 * abs(an) + 2e-4, (k, fd, an)


def test_goal_masking_draws_the_reference_bernoulli_stream():
    """mask_cond (mdtv_transformer.py:302-310): in train() mode goals * (1 - bernoulli(goal_drop)), elementwise."""
    from mdt_policy_amd import configs
    from mdt_policy_amd.models.networks.mdtv_transformer import MDTVTransformer
    cfg = {k: v for k, v in configs.mdtv_tiny(goal_drop=0.3).items() if k != "_target_"}
    net = MDTVTransformer(**cfg)
    goal = torch.from_numpy(synthetic.normal("goal", (7, 1, 512), 5))
    net.train()
    torch.manual_seed(11)
    got = net._goals(goal, False)
    torch.manual_seed(11)
    want = goal * (1.0 - torch.bernoulli(torch.ones((7, 1, 512)) * 0.3))
    assert torch.equal(got, want) and 0.2 < float((got == 0).float().mean()) < 0.4
    net.eval()
    assert net._goals(goal, False) is goal
    assert torch.equal(net._goals(goal, True), torch.zeros_like(goal))


@pytest.mark.gpu
def test_hip_training_under_lightning_style_amp():
    """The reference trains with precision 16 (conf/config.yaml:46): autocast + GradScaler around the step.  The HIP
    path stays fp32 inside; half-precision inputs are cast on the way in (differentiably) and the scaled loss
    gradient flows through the backward, so the unscaled gradients equal the plain fp32 ones."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case("mdtv_tiny")
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    args = (li["actions"].cuda(), goal.cuda(), li["noise_train"].cuda(), li["sigma"].cuda())
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    loss, _ = model.loss(gstate, *args)
    loss.backward()
    ref = {k: p.grad.clone() for k, p in model.inner_model.named_parameters() if p.grad is not None}
    model.zero_grad()
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    tokens16 = gstate["state_images"].half().requires_grad_()           # e.g. an upstream module that ran in fp16
    with torch.autocast("cuda", dtype=torch.float16):
        l16, _ = model.loss(dict(gstate, state_images=tokens16), *args)
    assert l16.dtype == torch.float32 and abs(l16.item() - loss.item()) <= 2e-3 * abs(loss.item())
    scaler.scale(l16).backward()
    assert tokens16.grad is not None and tokens16.grad.dtype == torch.float16
    scaler.unscale_(opt)
    for k, p in model.inner_model.named_parameters():
        if k in ref:
            assert_close(p.grad.cpu(), ref[k].cpu(), rtol=5e-3, atol=5e-3 * float(ref[k].abs().max()) + 1e-7, what=k)
    scaler.step(opt); scaler.update()


@pytest.mark.gpu
def test_context_only_uncond_under_autograd():
    """forward_context_only(..., uncond=True) in a training step (classifier-free-guidance style auxiliary passes): the goal is
    zeroed before the goal embedder (mdtv_transformer.py:256-257) and the call stays differentiable."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, state, goal, li = case("mdtv_tiny")
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    gstate = {k: (v.cuda().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    ctx = model.forward_context_only(gstate, None, goal.cuda(), li["sigma"].cuda(), uncond=True)
    w = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), 9))
    (ctx * w.cuda()).sum().backward()
    P = {k: v.double().requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
    st64 = {k: (v.double().requires_grad_() if torch.is_tensor(v) else v) for k, v in state.items()}
    c64 = O.encode(P, cfg, st64, torch.zeros_like(goal).double(), "mdtv", "enc_only")
    (c64 * w.double()).sum().backward()
    assert_close(ctx.detach().cpu(), c64.detach(), what="unconditional context")
    assert_close(gstate["state_images"].grad.cpu(), st64["state_images"].grad, rtol=2e-3, atol=1e-5, what="d_state_images")
    ref = P["inner_model.tok_emb.weight"].grad
    assert_close(model.inner_model.tok_emb.weight.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()), what="tok_emb")


@pytest.mark.gpu
def test_tape_reuse_across_streams_waits_for_the_previous_backward():
    """A tape released at the end of a backward on stream A and picked up by a forward on stream B: mdt_tape_release leaves
    an event on A that the next user waits for, so the new forward cannot overwrite activations the in-flight backward
    still reads.  Alternating streams, every step's gradients must equal the single-stream result bit for bit."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx, cfg, _, _, _ = case("mdtv_default")
    B = 256
    state, goal, _ = inputs_of(meta, batch=B)
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, meta["loss_seed"]).items()}
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    ggoal = goal.cuda()

    def step():
        for p in model.parameters():
            p.grad = None
        loss, _ = model.loss(gstate, li["actions"], ggoal, li["noise_train"], li["sigma"])
        loss.backward()
        return torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone()

    want = step()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    gots = []
    for i in range(6):
        s = streams[i % 2]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gots.append(step())
        # no synchronisation between iterations: the next forward is enqueued while this backward may still run
    torch.cuda.synchronize()
    for i, got in enumerate(gots):
        assert torch.equal(got, want), f"step {i}"


@pytest.mark.gpu
def test_library_allocations_succeed_while_torch_caches_the_free_memory():
    """Under a real training run torch's caching allocator holds most of the GPU as cached-but-free blocks (the ResNet / CLIP
    encoders' activations) and HIP reports almost nothing free.  The library's batch-sized buffers (workspace, tapes, backward
    scratch) therefore come out of torch's pool (utils/torch_allocator.py over mdt_set_allocator): here torch caches all but
    2 GiB of the device as ONE block -- partly in use once the step starts, so `empty_cache()` could not release it for a raw
    hipMalloc -- and a B = 4096 training step still finds its several GiB of tape in it."""
    import os
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("squeezes the whole GPU: not next to other workers' tests (runs in a serial `pytest -m gpu`)")
    from mdt_policy_amd import configs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    cfg = configs.mdtv_default()
    B = 4096
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda().eval()
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    with torch.no_grad():
        model({"state_images": inp["state_images"][:2], "modality": "lang"}, li["actions"][:2], inp["goal"][:2],
              li["sigma"][:2])  # handle + weight arena exist before the squeeze
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    keep = 2 << 30
    if free < (24 << 30):
        pytest.skip("needs a mostly free GPU")
    hog = torch.empty(free - keep, dtype=torch.uint8, device="cuda")
    del hog  # the block stays in torch's cache: HIP sees ~2 GiB free
    assert torch.cuda.mem_get_info()[0] < keep + (1 << 30)
    allocated = torch.cuda.memory_allocated()
    loss, _ = model.loss({k: v for k, v in state.items()}, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    grown = torch.cuda.memory_allocated() - allocated
    assert grown > keep, f"the library's buffers are not in torch's pool (memory_allocated grew by {grown >> 20} MiB only)"


@pytest.mark.gpu
@pytest.mark.parametrize("B,mode", [(37, "train"), (128, "eval"), (256, "train")])
def test_training_step_with_its_side_stream_is_bit_reproducible(B, mode):
    """Round 6: the blocks' weight gradients, the decoder tail and the forward's context-independent preparation run on a side
    stream beside the chain.  Every kernel is deterministic, so the same seeded step must leave the same BITS in the loss and in
    every gradient every time -- a difference is a race between the streams (tools/train_soak.py runs 1200 such steps)."""
    from mdt_policy_amd import configs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    cfg = configs.mdtv_default()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda()
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"].requires_grad_(), "modality": "lang"}
    goal = inp["goal"].requires_grad_()
    model.train(mode == "train")
    ref = None
    for it in range(25):
        torch.manual_seed(1234)
        model.zero_grad(set_to_none=True)
        state["state_images"].grad = None
        goal.grad = None
        loss, _ = model.loss(state, li["actions"], goal, li["noise_train"], li["sigma"])
        (loss + 0.1 * model.inner_model.latent_encoder_emb.square().mean()).backward()
        got = [loss.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None] + \
              [state["state_images"].grad.clone(), goal.grad.clone()]
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), f"step {it} differs from the first one"


@pytest.mark.gpu
def test_training_step_with_the_bf16_split_products_equals_the_fp32_products_to_fp32_rounding():
    """Round 6: at B = 1024 the K = 384 products of the step (12 288-row d x d, qkv, c_fc and c_fc-gradient GEMMs) run the
    weight-stationary body in its three-way bf16 split form (mdt_ws.h).  The split keeps fp32's product accuracy, not its bits:
    the same seeded step (eval mode: no dropout) with the split on and off must agree in the loss and in EVERY gradient to a few
    fp32 roundings of the gradient's own size -- tolerance 2e-5 of max |g| per tensor, the level at which two fp32 summation
    orders of these 12 288-row reductions differ -- and the split must really have run (bits differ somewhere)."""
    from mdt_policy_amd import _lib, configs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    L = _lib.load()
    B = 1024
    cfg = configs.mdtv_default()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda().eval()
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"].requires_grad_(), "modality": "lang"}
    goal = inp["goal"].requires_grad_()
    res = {}
    try:
        for split in (0, 1):
            L.mdt_op_set_ws_split(split)
            model.zero_grad(set_to_none=True)
            state["state_images"].grad = None
            goal.grad = None
            loss, _ = model.loss(state, li["actions"], goal, li["noise_train"], li["sigma"])
            loss.backward()
            torch.cuda.synchronize()
            res[split] = [("loss", loss.detach().clone())] + \
                         [(n, p.grad.clone()) for n, p in model.named_parameters() if p.grad is not None] + \
                         [("d state_images", state["state_images"].grad.clone()), ("d goal", goal.grad.clone())]
    finally:
        L.mdt_op_set_ws_split(-1)
    assert len(res[0]) == len(res[1]) > 100
    differs = False
    # (gradients that are zero in exact arithmetic -- the key biases: softmax ignores a per-query constant -- are rounding noise of
    # the tensors around them: the floor is taken from the largest parameter gradient of the step)
    floor = 1e-7 * max(a.abs().max().item() for n, a in res[0] if n.startswith("inner_model."))
    worst = 0.0
    for (n, a), (_, b) in zip(res[0], res[1]):
        differs |= not torch.equal(a, b)
        scale = a.abs().max().item()
        err = (a - b).abs().max().item()
        worst = max(worst, err / (scale + floor / 2e-5))
        assert err <= 2e-5 * scale + floor, f"{n}: split and fp32 products differ by {err:.3g} at max |g| = {scale:.3g}"
    print(f"worst |split - fp32| / max |g| over {len(res[0])} tensors: {worst:.3g}")
    assert differs, "the split form did not run: every gradient has the fp32 products' bits"


@pytest.mark.gpu
def test_large_batch_sampling_after_optimizer_steps_reads_fresh_split_images():
    """Round 6: large-batch sampling (>= 768 rows a launch) multiplies pre-split bf16 images of the MLP / qkv weights.  While a training
    state exists, parameter loads do not rewrite those images (every optimizer step would); they are marked stale and re-made from the
    fp32 fragment images by the next call that reads them.  After two optimizer steps a B = 128 sampler call must give the BITS of a
    fresh model that loaded the same weights (whose images come from the raw weights), differ from the fp32 launches' bits (the split
    form ran) and agree with them to fp32 rounding."""
    from mdt_policy_amd import _lib, configs
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    L = _lib.load()
    cfg = configs.mdtv_default()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda()
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}, strict=False)
    sig = gs.get_sigmas_exponential(5, 0.001, 80.0).cuda()
    big = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(128, cfg, 3).items()}
    st_big = {"state_images": big["state_images"], "modality": "lang"}
    L.mdt_op_set_mlp_split(1)   # (whatever MDT_HIP_MLP_SPLIT says: this test is about the split images)
    with torch.no_grad():
        model.eval()
        before = model.sample_ddim(st_big, big["noise"] * 80.0, big["goal"], sig).clone()   # images of the initial weights in use
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    tr = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(64, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(64, cfg, 2).items()}
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss({"state_images": tr["state_images"], "modality": "lang"}, li["actions"], tr["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        opt.step()
    model.eval()
    with torch.no_grad():
        got = model.sample_ddim(st_big, big["noise"] * 80.0, big["goal"], sig).clone()
        fresh = GCDenoiser(cfg, 0.5).cuda().eval()
        fresh.load_state_dict(model.state_dict())
        try:
            want = fresh.sample_ddim(st_big, big["noise"] * 80.0, big["goal"], sig).clone()
            L.mdt_op_set_mlp_split(0)
            fp32 = fresh.sample_ddim(st_big, big["noise"] * 80.0, big["goal"], sig).clone()
        finally:
            L.mdt_op_set_mlp_split(-1)
    torch.cuda.synchronize()
    assert not torch.equal(got, before), "the optimizer steps did not change the sampler's output"
    assert torch.equal(got, want), "sampling after optimizer steps read stale split images"
    assert not torch.equal(want, fp32), "the split launches did not run at B = 128"
    assert_close(want.cpu(), fp32.cpu(), rtol=1e-4, atol=1e-4, what="split against fp32 launches, B = 128 sampler call")


@pytest.mark.gpu
def test_graph_replayed_large_batch_sampling_after_optimizer_steps_is_recaptured(monkeypatch):
    """The same with the sampler call replayed from a HIP graph (MDT_HIP_GRAPH=1 forces the replay at any batch): a graph captured
    before the optimizer steps holds the split launches but not the refresh of their images -- the library moves the generation
    the graph owner compares when a load leaves the images stale, so the call is captured again and reads fresh images."""
    from mdt_policy_amd import _lib, configs
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    L = _lib.load()
    monkeypatch.setattr(gs, "_GRAPH_MODE", "1", raising=False)
    monkeypatch.setattr(gs, "_GRAPH_SAMPLER", True, raising=False)
    cfg = configs.mdtv_default()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda()
    sig = gs.get_sigmas_exponential(4, 0.001, 80.0).cuda()
    big = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(100, cfg, 3).items()}
    st_big = {"state_images": big["state_images"], "modality": "lang"}
    L.mdt_op_set_mlp_split(1)
    try:
        model.eval()
        with torch.no_grad():
            for _ in range(4):   # capture + replays on the initial weights
                before = gs.sample_ddim(model, st_big, big["noise"] * 80.0, big["goal"], sig).clone()
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        tr = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(32, cfg, 1).items()}
        li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(32, cfg, 2).items()}
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss, _ = model.loss({"state_images": tr["state_images"], "modality": "lang"}, li["actions"], tr["goal"], li["noise_train"], li["sigma"])
            loss.backward()
            opt.step()
        model.eval()
        with torch.no_grad():
            got = [gs.sample_ddim(model, st_big, big["noise"] * 80.0, big["goal"], sig).clone() for _ in range(3)]
            fresh = GCDenoiser(cfg, 0.5).cuda().eval()
            fresh.load_state_dict(model.state_dict())
            want = fresh.sample_ddim(st_big, big["noise"] * 80.0, big["goal"], sig).clone()
    finally:
        L.mdt_op_set_mlp_split(-1)
    torch.cuda.synchronize()
    assert getattr(model, "_graphed_samplers", None), "the replay path was not taken"
    assert not torch.equal(got[0], before), "the optimizer steps did not change the sampler's output"
    for i, g in enumerate(got):
        assert torch.equal(g, want), f"replayed call {i} after the optimizer steps read stale split images"
