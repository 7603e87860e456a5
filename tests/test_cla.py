"""Contrastive (CLA) head, SURVEY.md 8(f) item 4: ClipStyleProjection / MAPBlock and the InfoNCE loss.
CPU: the oracle against outputs and gradients of the REFERENCE (tests/golden/g14_cla_*.npz); the facade's loss function
against the reference's clip_auxiliary_loss; the differentiable all-gather on two gloo processes.
GPU: the HIP MAPBlock (forward, taped forward, backward) against both."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from mdt_policy_amd import synthetic
from oracle import cla_oracle as O
from tests.helpers import assert_close, load_fixture

STYLES = ["map_tiny", "map_default", "map_state_only", "map_five_tokens", "mean_pooling", "mlp", "single_token"]


def fixture(name):
    meta, fx = load_fixture(f"g14_cla_{name}.npz")
    kw = meta["kwargs"]
    P = {k: torch.from_numpy(v) for k, v in
         synthetic.fill_state_dict([(k, tuple(s)) for k, s in meta["state_dict"]], meta["weight_seed"], meta["profile"]).items()}
    x = torch.from_numpy(synthetic.normal("ctx", (meta["B"], meta["N"], kw["token_dim"]), meta["input_seed"]))
    cot = torch.from_numpy(synthetic.normal("cotangent", tuple(fx["out"].shape), meta["cot_seed"]))
    return meta, fx, kw, P, x, cot


def summary(g):
    g = g.detach().double().cpu()
    return [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]


def check_grads(got, want, what, rtol=2e-3):
    assert set(got) == set(want), what
    for k, w in want.items():
        g, w = np.array(got[k]), np.array(w)
        tol = rtol * abs(w[0]) + 1e-6
        assert np.all(np.abs(g - w) <= tol), f"{what} {k}: {g} vs {w}"


@pytest.mark.parametrize("name", STYLES)
def test_oracle_matches_the_reference_projection_and_its_gradients(name):
    meta, fx, kw, P, x, cot = fixture(name)
    P = {k: v.double().requires_grad_() for k, v in P.items()}
    x = x.double().requires_grad_()
    out = O.clip_style_projection(P, x, kw["clip_style"], kw.get("clip_token_index", 0))
    assert_close(out.detach(), fx["out"], rtol=1e-4, atol=1e-5, what="projection")
    (out * cot.double()).sum().backward()
    check_grads({k: summary(v.grad) for k, v in P.items()}, meta["grads"], name)
    assert_close(x.grad, fx["d_x"], rtol=1e-3, atol=1e-6, what="d_x")


def _infonce_cases():
    meta, fx = load_fixture("g14_cla_infonce.npz")
    for c in meta["cases"]:
        img = torch.from_numpy(synthetic.normal("img", (c["B"], c["D"]), 145))
        lang = torch.from_numpy(synthetic.normal("lang", (c["B"], c["D"]), 146) + 0.5 * synthetic.normal("img", (c["B"], c["D"]), 145))
        yield c, fx, img, lang


def _check_infonce(fn, device, rtol, tight):
    n = 0
    for c, fx, img, lang in _infonce_cases():
        img, lang = img.to(device).requires_grad_(), lang.to(device).requires_grad_()
        ls = torch.tensor(c["logit_scale"], dtype=torch.float32, device=device, requires_grad=True)
        loss = fn(img, lang, ls, mode=c["mode"])
        (3.0 * loss).backward()
        k = c["key"]
        assert_close(loss.detach().reshape(1).cpu(), fx[k + "_loss"].reshape(1), rtol=rtol, atol=1e-6 if tight else 1e-4, what=k)
        for got, name in ((img.grad, "_d_img"), (lang.grad, "_d_lang"), (ls.grad.reshape(1), "_d_scale")):
            want = 3.0 * fx[k + name].reshape(got.shape)
            assert_close(got.cpu(), want, rtol=rtol, atol=(1e-7 if tight else 1e-3 * float(np.abs(want).max()) + 1e-7),
                         what=k + name)
        n += 1
    assert n == 9
    with pytest.raises(ValueError):
        fn(img, lang, ls, mode="both")


def test_oracle_infonce_matches_the_reference_loss_and_gradients():
    _check_infonce(O.clip_auxiliary_loss, "cpu", 1e-4, True)


def test_infonce_has_no_cpu_path():
    from mdt_policy_amd.models.contrastive import clip_auxiliary_loss
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        clip_auxiliary_loss(torch.zeros(4, 16), torch.zeros(4, 16), torch.tensor(0.0))


def test_facade_state_dict_matches_the_reference_names_and_order():
    from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
    for name in STYLES:
        meta, _ = load_fixture(f"g14_cla_{name}.npz")
        m = ClipStyleProjection(**meta["kwargs"])
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == meta["state_dict"], name
        assert [k for k, _ in m.named_parameters()] == [k for k, _ in meta["state_dict"]]
    with pytest.raises(ValueError):
        ClipStyleProjection("nope")


def test_map_block_has_no_cpu_path():
    from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
    m = ClipStyleProjection("map", 128, 1, 4)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m(torch.zeros(2, 4, 128))


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    from mdt_policy_amd.models.contrastive import all_gather_with_grad
    clip_auxiliary_loss = O.clip_auxiliary_loss  # CPU processes: the loss itself is the oracle's, the gather the facade's
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    B, D = 3, 16
    img = torch.from_numpy(synthetic.normal("img", (world * B, D), 7))[rank * B:(rank + 1) * B].clone().requires_grad_()
    lang = torch.from_numpy(synthetic.normal("lang", (world * B, D), 8))[rank * B:(rank + 1) * B].clone().requires_grad_()
    ls = torch.tensor(1.0, requires_grad=True)
    loss = clip_auxiliary_loss(all_gather_with_grad(img).flatten(0, 1), all_gather_with_grad(lang).flatten(0, 1), ls)
    loss.backward()
    q.put((rank, loss.item(), img.grad.numpy(), lang.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_with_grad_reproduces_the_global_batch_loss():
    """Two gloo processes, each with half of the batch: the loss equals the single-process loss on the whole batch and
    every rank's gradient is the matching slice of the whole-batch gradient times the number of ranks (each rank
    back-propagates its own copy of the global loss and the gathers sum them -- exactly what DDP then averages)."""
    clip_auxiliary_loss = O.clip_auxiliary_loss
    from tests.test_sharding_gloo import _free_port as free_port
    world, B, D = 2, 3, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    img = torch.from_numpy(synthetic.normal("img", (world * B, D), 7)).requires_grad_()
    lang = torch.from_numpy(synthetic.normal("lang", (world * B, D), 8)).requires_grad_()
    loss = clip_auxiliary_loss(img, lang, torch.tensor(1.0))
    loss.backward()
    for rank, l, gi, gl in res:
        assert abs(l - loss.item()) < 1e-6
        assert_close(torch.from_numpy(gi), world * img.grad[rank * B:(rank + 1) * B], rtol=1e-5, atol=1e-7, what="d_img")
        assert_close(torch.from_numpy(gl), world * lang.grad[rank * B:(rank + 1) * B], rtol=1e-5, atol=1e-7, what="d_lang")


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_infonce_matches_the_reference_loss_and_gradients():
    from mdt_policy_amd.models.contrastive import clip_auxiliary_loss
    _check_infonce(clip_auxiliary_loss, "cuda", 1e-3, False)


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,mode", [(1024, 384, "symmetric"), (2048, 384, "symmetric"), (333, 128, "text_to_img"),
                                      (17, 64, "img_to_text")])
def test_hip_infonce_large_batches_against_float64(B, D, mode):
    """The global batch of the reference's training (8 x 128 ... 2048) and ragged sizes, against float64 autograd."""
    from mdt_policy_amd.models.contrastive import clip_auxiliary_loss
    img = torch.from_numpy(synthetic.normal("img", (B, D), 21))
    lang = torch.from_numpy(0.7 * synthetic.normal("img", (B, D), 21) + synthetic.normal("lang", (B, D), 22))
    i64, l64 = img.double().requires_grad_(), lang.double().requires_grad_()
    s64 = torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float64, requires_grad=True)
    ref = O.clip_auxiliary_loss(i64, l64, s64, mode)
    ref.backward()
    ig, lg = img.cuda().requires_grad_(), lang.cuda().requires_grad_()
    sg = torch.tensor(float(np.log(1 / 0.07)), device="cuda", requires_grad=True)
    loss = clip_auxiliary_loss(ig, lg, sg, mode)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item()) + 1e-6
    assert abs(sg.grad.item() - s64.grad.item()) <= 1e-3 * abs(s64.grad.item()) + 1e-6
    assert_close(ig.grad.cpu(), i64.grad, rtol=1e-3, atol=1e-3 * float(i64.grad.abs().max()), what="d_img")
    assert_close(lg.grad.cpu(), l64.grad, rtol=1e-3, atol=1e-3 * float(l64.grad.abs().max()), what="d_lang")
    with torch.no_grad():  # value only: no gradient buffers are produced
        assert abs(clip_auxiliary_loss(ig, lg, sg, mode).item() - loss.item()) <= 1e-6 * abs(loss.item())

@pytest.mark.gpu
@pytest.mark.parametrize("name", STYLES)
def test_hip_projection_matches_reference_and_oracle(name):
    from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
    meta, fx, kw, P, x, cot = fixture(name)
    m = ClipStyleProjection(**kw)
    m.load_state_dict(P)
    m = m.cuda()
    with torch.no_grad():
        out = m(x.cuda())
    assert_close(out.cpu(), fx["out"], rtol=1e-3, atol=1e-4, what="inference forward")
    xg = x.cuda().requires_grad_()
    out = m(xg)
    assert_close(out.detach().cpu(), fx["out"], rtol=1e-3, atol=1e-4, what="taped forward")
    (out * cot.cuda()).sum().backward()
    check_grads({k: summary(p.grad) for k, p in m.named_parameters()}, meta["grads"], name + " vs reference")
    assert_close(xg.grad.cpu(), fx["d_x"], rtol=2e-3, atol=2e-3 * float(np.abs(fx["d_x"]).max()), what="d_x")
    if not P:  # mean pooling / single token: nothing to train
        return
    # full tensors against float64 autograd through the oracle
    P64 = {k: v.double().requires_grad_() for k, v in P.items()}
    o64 = O.clip_style_projection(P64, x.double(), kw["clip_style"], kw.get("clip_token_index", 0))
    (o64 * cot.double()).sum().backward()
    for k, p in m.named_parameters():
        ref = P64[k].grad
        assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-7, what=k)


@pytest.mark.gpu
def test_hip_map_block_two_tapes_batches_and_tape_rules():
    """The agent pools two contexts before ONE backward; batch sizes change between calls."""
    from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
    meta, fx, kw, P, x, cot = fixture("map_tiny")
    m = ClipStyleProjection(**kw)
    m.load_state_dict(P)
    m = m.cuda()
    xa = x.cuda().requires_grad_()
    xb = torch.from_numpy(synthetic.normal("ctx_b", (9, 4, 128), 5)).cuda().requires_grad_()
    oa, ob = m(xa), m(xb)
    (oa.sum() + (ob ** 2).sum()).backward()
    P64 = {k: v.double().requires_grad_() for k, v in P.items()}
    xa64, xb64 = x.double().requires_grad_(), xb.detach().cpu().double().requires_grad_()
    ra, rb = O.clip_style_projection(P64, xa64, "map"), O.clip_style_projection(P64, xb64, "map")
    (ra.sum() + (rb ** 2).sum()).backward()
    for k, p in m.named_parameters():
        ref = P64[k].grad
        assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-7, what=k)
    assert_close(xb.grad.cpu(), xb64.grad, rtol=2e-3, atol=1e-5, what="d_x second tape")
    # a larger batch afterwards (workspace / tape growth), then the small one again
    big = torch.from_numpy(synthetic.normal("ctx_c", (300, 4, 128), 6))
    with torch.no_grad():
        ob2 = m(big.cuda())
    assert_close(ob2.cpu(), O.clip_style_projection(P, big, "map"), rtol=1e-3, atol=1e-4, what="B=300")
    with torch.no_grad():
        assert_close(m(x.cuda()).cpu(), fx["out"], rtol=1e-3, atol=1e-4, what="small batch again")
    # an optimizer step is picked up (parameters are re-uploaded when their version changes)
    with torch.no_grad():
        m.latent_proj.attn_norm.g.mul_(1.5)
        P2 = dict(P, **{"latent_proj.attn_norm.g": P["latent_proj.attn_norm.g"] * 1.5})
        assert_close(m(x.cuda()).cpu(), O.clip_style_projection(P2, x, "map"), rtol=1e-3, atol=1e-4, what="after update")


@pytest.mark.gpu
def test_hip_contrastive_loss_end_to_end_against_the_oracles():
    """compute_contrastive_loss on the HIP denoiser + HIP MAPBlock: value and gradients (goal embedders, encoder,
    pooling head, temperature) against float64 autograd through the two oracles."""
    from mdt_policy_amd import configs
    from mdt_policy_amd.models.contrastive import compute_contrastive_loss
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.models.networks.transformers.transformer_blocks import ClipStyleProjection
    from oracle import mdt_oracle as MO
    from tests.helpers import params_of
    meta, _ = load_fixture("g11_grads_mdtv_tiny.npz")
    # clip_extra_forward puts the model into train() mode as the reference does: no dropout here, so that the
    # float64 oracles (eval arithmetic) are the exact expectation
    cfg = configs.mdtv_tiny(attn_pdrop=0.0, resid_pdrop=0.0, mlp_pdrop=0.0)
    B = 6
    model = GCDenoiser(cfg, 0.5)
    PM = params_of(meta)
    model.load_state_dict(PM)
    model = model.cuda().eval()
    cmeta, _, kw, PC, _, _ = fixture("map_tiny")
    clip = ClipStyleProjection(**kw)
    clip.load_state_dict(PC)
    clip = clip.cuda()
    ls = torch.nn.Parameter(torch.tensor(float(np.log(1 / 0.07)), device="cuda"))
    inp = synthetic.sampler_inputs(B, cfg, 31, "mdtv")
    li = synthetic.loss_inputs(B, cfg, 32)
    img_goal = torch.from_numpy(synthetic.normal("img_goal", (B, 1, 512), 33))
    state = {"state_images": torch.from_numpy(inp["state_images"]).cuda(), "modality": "lang"}
    a, nz, sg = (torch.from_numpy(li[k]).cuda() for k in ("actions", "noise_train", "sigma"))
    loss, _ = model.loss(state, a, torch.from_numpy(inp["goal"]).cuda(), nz, sg)
    vis_state = dict(state, modality="vis")
    cont = compute_contrastive_loss(model, clip, ls, vis_state, img_goal.cuda(), a, sg, nz)
    (loss + cont).backward()
    # float64 oracles
    P = {k: v.double().requires_grad_(v.dtype.is_floating_point) for k, v in PM.items()}
    Q = {k: v.double().requires_grad_() for k, v in PC.items()}
    st = {"state_images": torch.from_numpy(inp["state_images"]).double(), "modality": "lang"}
    l64, _ = MO.loss(P, cfg, st, torch.from_numpy(li["actions"]).double(), torch.from_numpy(inp["goal"]).double(),
                     torch.from_numpy(li["noise_train"]).double(), torch.from_numpy(li["sigma"]).double())
    ctx_l = MO.encode(P, cfg, st, torch.from_numpy(inp["goal"]).double(), "mdtv", "forward")
    ctx_v = MO.forward_context_only(P, cfg, dict(st, modality="vis"), img_goal.double())
    ls64 = torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float64, requires_grad=True)
    c64 = O.clip_auxiliary_loss(O.clip_style_projection(Q, ctx_v, "map"), O.clip_style_projection(Q, ctx_l, "map"), ls64)
    (l64 + c64).backward()
    assert abs(cont.item() - c64.item()) <= 1e-3 * abs(c64.item())
    assert abs(ls.grad.item() - ls64.grad.item()) <= 2e-3 * abs(ls64.grad.item()) + 1e-6
    for k, p in clip.named_parameters():
        ref = Q[k].grad
        assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-7, what="clip " + k)
    for k in ("goal_emb.0.weight", "lang_emb.2.weight", "tok_emb.weight", "encoder.blocks.0.attn.key.weight", "encoder.ln.weight",
              "decoder.blocks.1.mlp.c_fc.weight"):
        ref = P["inner_model." + k].grad
        got = dict(model.inner_model.named_parameters())[k].grad
        assert_close(got.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-7, what=k)
