"""Multi-tensor optimizer kernels (pytest -m gpu): FusedAdamW against torch.optim.AdamW, the EMA update against the
reference callback's arithmetic (mdt/callbacks/ema.py:84-126)."""
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(384, 384), (1536,), (7, 384), (3, 5, 7), (1,), (4097,), (9216, 384)]
    return [torch.randn(s, generator=g) for s in shapes]


def test_fused_adamw_matches_torch_adamw_over_steps_and_groups():
    from mdt_policy_amd.optim import FusedAdamW
    base = _params(0)
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    groups = lambda ps: [{"params": ps[:4], "weight_decay": 0.05}, {"params": ps[4:], "weight_decay": 0.0, "lr": 3e-4}]
    oa = FusedAdamW(groups(pa), lr=1e-3, betas=(0.9, 0.95))
    ob = torch.optim.AdamW(groups(pb), lr=1e-3, betas=(0.9, 0.95))
    for step in range(6):
        grads = _params(100 + step)
        for i, (a, b, g) in enumerate(zip(pa, pb, grads)):
            if step == 2 and i == 1:       # a parameter without gradient this step is skipped (and keeps its step count)
                a.grad = b.grad = None
                continue
            a.grad, b.grad = g.cuda(), g.cuda().clone()
        va = [p._version for p in pa]
        oa.step(); ob.step()
        assert all(p._version > v for p, v, g in zip(pa, va, pa) if p.grad is not None)
    for a, b in zip(pa, pb):
        assert_close(a.detach().cpu(), b.detach().cpu(), rtol=2e-6, atol=2e-7, what="parameter")
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sb:
        assert float(sa[k]["step"]) == float(sb[k]["step"])
        assert_close(sa[k]["exp_avg"].cpu(), sb[k]["exp_avg"].cpu(), rtol=2e-6, atol=5e-7, what="exp_avg")  # torch forms m by lerp: last-ulp differences
        assert_close(sa[k]["exp_avg_sq"].cpu(), sb[k]["exp_avg_sq"].cpu(), rtol=5e-6, atol=1e-9, what="exp_avg_sq")
    ob2 = torch.optim.AdamW(groups(pb), lr=1e-3, betas=(0.9, 0.95))
    ob2.load_state_dict(oa.state_dict())     # the state layout is torch's


def test_fused_adamw_refuses_cpu_parameters():
    from mdt_policy_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        FusedAdamW([p]).step()


def test_multi_tensor_ema_matches_the_callback_loop():
    from mdt_policy_amd.callbacks.ema import get_decay, multi_tensor_ema
    ws = [t.cuda() for t in _params(1)] + [torch.arange(5, device="cuda")]
    ema = [t.cuda() for t in _params(2)] + [torch.zeros(5, dtype=torch.int64, device="cuda")]
    ref = [e.clone() for e in ema]
    for step in (1, 2, 10, 1000):
        d = get_decay(step)
        assert d == max(min(1 - (1 + max(0, step - 1) / 1.0) ** -(2 / 3), 0.9999), 0.0)
        multi_tensor_ema(ema, ws, d)
        for e, w in zip(ref, ws):           # EMA.apply_ema (reference ema.py:117-126)
            if w.dtype == torch.int64:
                e.data = w.data.clone()
            else:
                diff = e.data - w.data
                diff.mul_(1.0 - d)
                e.sub_(diff)
    for a, b in zip(ema, ref):
        assert_close(a.cpu().double(), b.cpu().double(), rtol=1e-6, atol=1e-7, what="ema")
    assert get_decay(0) == 0.0 and get_decay(10 ** 9) == 0.9999


def test_training_with_fused_adamw_reaches_the_kernels():
    """FusedAdamW writes the parameters in place from the library: the denoiser must pick the new weights up."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.optim import FusedAdamW
    torch.manual_seed(0)
    cfg = configs.mdtv_tiny()
    model = GCDenoiser(cfg, 0.5).cuda().eval()
    B = 8
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"], "modality": "vis"}
    opt = FusedAdamW(model.parameters(), lr=2e-3, weight_decay=0.05)
    losses = []
    for _ in range(10):
        opt.zero_grad()
        loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.gpu
def test_torch_fused_optimizer_steps_are_seen_by_the_weight_cache():
    """torch's fused optimizers update parameters WITHOUT bumping the autograd version counter the weight cache keys on;
    the optimizer post-step hook (mdt_policy_amd/utils/weight_cache.py) marks the module dirty instead.  After a fused step
    the HIP forward must agree with the oracle evaluated on the UPDATED parameters."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from oracle import mdt_oracle as O
    from tests.helpers import assert_close
    cfg = configs.mdtv_tiny()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda().eval()
    B = 4
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    opt = torch.optim.AdamW(model.parameters(), lr=3e-2, fused=True)
    p0 = model.inner_model.tok_emb.weight
    v0 = p0._version
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        before = p0.detach().clone()
        opt.step()
        assert not torch.equal(before, p0.detach())
    if p0._version != v0:
        pytest.skip("this torch build bumps version counters in fused optimizer steps: nothing to guard")
    x = li["actions"] + li["noise_train"] * li["sigma"][:, None, None]
    with torch.no_grad():
        got = model(state, x, inp["goal"], li["sigma"])
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    st = {"state_images": inp["state_images"].cpu(), "modality": "lang"}
    want = O.denoise(P, cfg, st, x.cpu(), inp["goal"].cpu(), li["sigma"].cpu(), 0.5, "mdtv")
    assert_close(got.cpu(), want, what="forward after two fused optimizer steps")


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["deepcopy", "pickle"])
def test_fused_optimizer_steps_on_a_copy_of_the_module_are_seen(how):
    """A module made by copy.deepcopy / pickle never ran __init__: it registers with the optimizer hook in __setstate__.
    Two fused-AdamW steps on the COPY (after the copy has been used once, so that it owns uploaded images), then the HIP
    forward of the copy against the oracle on the copy's updated parameters."""
    import copy
    import pickle
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from oracle import mdt_oracle as O
    from tests.helpers import assert_close
    cfg = configs.mdtv_tiny()
    torch.manual_seed(0)
    orig = GCDenoiser(cfg, 0.5).cuda().eval()
    model = copy.deepcopy(orig) if how == "deepcopy" else pickle.loads(pickle.dumps(orig))
    B = 4
    inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    x = li["actions"] + li["noise_train"] * li["sigma"][:, None, None]
    with torch.no_grad():
        first = model(state, x, inp["goal"], li["sigma"]).clone()  # uploads the copy's images
    opt = torch.optim.AdamW(model.parameters(), lr=3e-2, fused=True)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(state, li["actions"], inp["goal"], li["noise_train"], li["sigma"])
        loss.backward()
        opt.step()
    with torch.no_grad():
        got = model(state, x, inp["goal"], li["sigma"])
        untouched = orig(state, x, inp["goal"], li["sigma"])
    assert not torch.equal(got, first)
    assert torch.equal(untouched, first), "stepping the copy changed the original"
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    st = {"state_images": inp["state_images"].cpu(), "modality": "lang"}
    want = O.denoise(P, cfg, st, x.cpu(), inp["goal"].cpu(), li["sigma"].cpu(), 0.5, "mdtv")
    assert_close(got.cpu(), want, what=f"forward of the {how} copy after two fused optimizer steps")


@pytest.mark.gpu
def test_five_training_steps_follow_the_oracle_trajectory():
    """End to end: five optimizer steps of the HIP path (loss forward, HIP backward, FusedAdamW, batched re-upload) against the
    same five steps taken on the CPU with float64 autograd through the oracle and torch.optim.AdamW -- the loss of every step
    and every parameter after the last one."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.optim import FusedAdamW
    from oracle import mdt_oracle as O
    cfg = configs.mdtv_tiny()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5)
    P = {k: v.detach().clone().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    B = 6
    inp = {k: torch.from_numpy(v) for k, v in synthetic.sampler_inputs(B, cfg, 1).items()}
    li = {k: torch.from_numpy(v) for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    state = {"state_images": inp["state_images"].cuda(), "modality": "lang"}
    st64 = {"state_images": inp["state_images"].double(), "modality": "lang"}
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
    trainable = [v for k, v in P.items() if v.requires_grad and not k.endswith("rotary_pos_emb.freqs")]
    ref_opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=0.05)
    for step in range(5):
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(state, li["actions"].cuda(), inp["goal"].cuda(), li["noise_train"].cuda(), li["sigma"].cuda())
        loss.backward()
        opt.step()
        ref_opt.zero_grad(set_to_none=True)
        l64, _ = O.loss(P, cfg, st64, li["actions"].double(), inp["goal"].double(), li["noise_train"].double(),
                        li["sigma"].double(), arch="mdtv")
        l64.backward()
        ref_opt.step()
        assert abs(loss.item() - l64.item()) <= 2e-3 * abs(l64.item()), f"step {step}: {loss.item()} vs {l64.item()}"
    got = model.state_dict()
    for k, v in P.items():
        if not v.requires_grad or v.grad is None:
            continue  # parameters the forward does not read (pos_emb, proprio_emb, the other goal embedder) only decay
        # Adam's update has size lr whatever the gradient's: an element whose gradient is rounding noise may step the other way
        # (5 steps x lr = 5e-3 at most); everything else must agree to fp32 accuracy
        d = (got[k].cpu().double() - v.detach()).abs()
        assert float(d.max()) <= 5.1e-3, k
        assert float((d > 2e-5 + 2e-3 * v.detach().abs()).double().mean()) <= 2e-3, f"{k}: too many elements off"
