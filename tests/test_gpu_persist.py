"""The persistent decoder kernel (mdt_persist.hip: the whole DDIM step loop as ONE launch, per-XCD sample ownership,
fence-free XCD-local barriers) against the launch sequence it replaces and against the reference's goldens.

Both paths run the same tile bodies (mdt_tiles.h) on the same tiles in the same k order, so their outputs must be
BIT-IDENTICAL; a stale read of another workgroup's activations inside the persistent kernel would show up as a
difference (and, in the soak, as a call that differs from the first one)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, load_fixture
from tests.test_gpu_parity import build, gpu_inputs, sampling

pytestmark = pytest.mark.gpu


def _lib():
    from mdt_policy_amd import _lib as L
    return L.load()


@pytest.fixture(autouse=True)
def _restore_switch(monkeypatch):
    # these tests count the persistent kernel's launches and compare launch paths: the automatic switch of repeated
    # rollout-sized calls to graph replay (gc_sampling._GRAPH_MODE "auto") stays off unless a test turns it on
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    monkeypatch.setattr(gs, "_GRAPH_MODE", "0")
    yield
    _lib().mdt_op_set_persist(-1)
    _lib().mdt_op_set_gemm_geometry(0)
    _lib().mdt_op_set_mlp_fuse_min(-1)
    _lib().mdt_op_set_attn_wide_min(-1)
    _lib().mdt_op_set_side_jobs(-1)


def _engine(model):
    return model.inner_model.hip_engine()


def _built():
    return bool(_lib().mdt_persist_built())


def _need_kernel():
    if not _built():
        pytest.skip("the shipped library carries the persistent kernel's entry points as stubs (build with "
                    "MDT_BUILD_PERSIST=1 to compare it with the launch sequence)")


def _sample(model, state, x_T, goal, sig, persist):
    if persist:
        _need_kernel()
    _lib().mdt_op_set_persist(1 if persist else 0)
    # the persistent kernel walks the two-GEMM MLP phases; large batches of the launch path fuse them into one launch that
    # adds the partial products in another order -- compare like with like
    _lib().mdt_op_set_mlp_fuse_min(-1 if persist else 0)
    _lib().mdt_op_set_attn_wide_min(-1 if persist else 0)  # likewise: attention in the projection's prologue sums in another order
    # ... and its small-batch geometry runs EVERY product on the split-K tiles, where the dispatcher now picks half-height tiles
    # for the wide ones: the launch path is held to the split-K kernel for the comparison
    _lib().mdt_op_set_gemm_geometry(-1 if (not persist and x_T.shape[0] <= 8) else 0)
    eng = _engine(model)
    n0 = eng.persist_launches()
    with torch.no_grad():
        out = sampling().sample_ddim(model, state, x_T, goal, sig)
    torch.cuda.synchronize()
    assert eng.persist_status() == 0
    used = eng.persist_launches() - n0
    assert used == (1 if persist else 0), f"persistent launches: {used}"
    return out


def test_unsupported_configuration_keeps_the_launch_sequence():
    """MDT (two state tokens + goal: Te = 3, H * Te = 24 with 8 heads of 64) at B = 8 has no instantiated variant of the
    SMALL kernel's attention (its 8 waves are 8 heads; fine) -- whatever the library decides, the call must succeed and
    match the reference; the switch must never make a call fail."""
    meta, fx = load_fixture("g3_b8_mdt.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    _lib().mdt_op_set_persist(1)
    with torch.no_grad():
        out = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, torch.from_numpy(fx["sigmas"]))
    torch.cuda.synchronize()
    assert _engine(model).persist_status() == 0
    assert_close(out.cpu(), fx["actions"], what="MDT default B=8")


@pytest.mark.parametrize("fixture,what", [("g3_b256_lang.npz", "MDT-V default B=256"), ("g3_b256_init.npz", "init weights B=256"),
                                          ("g1_tiny_mdtv.npz", "MDT-V tiny B=1"), ("g1_tiny_mdt.npz", "MDT tiny B=1")])
def test_persistent_matches_reference_and_launch_path(fixture, what):
    meta, fx = load_fixture(fixture)
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = torch.from_numpy(fx["sigmas"])
    x_T = noise * meta["sigma_max"]
    a = _sample(model, state, x_T, goal, sig, persist=True)
    b = _sample(model, state, x_T, goal, sig, persist=False)
    assert_close(a.cpu(), fx["actions"], what=f"{what}: persistent kernel vs reference")
    assert torch.equal(a, b), f"{what}: persistent kernel differs from the launch sequence, max |d| = {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8, 128, 136, 200, 250, 256, 300, 512])
def test_persistent_bit_identical_for_ragged_batches(B):
    """XCD x owns samples [x S, x S + S): the last XCDs hold fewer (or no) samples when B is not a multiple of 8."""
    meta, fx = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    from tests.helpers import inputs_of
    state, goal, noise = inputs_of(meta, batch=B)
    state = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    goal, noise = goal.cuda(), noise.cuda()
    sig = torch.from_numpy(fx["sigmas"])
    a = _sample(model, state, noise * 80.0, goal, sig, persist=True)
    b = _sample(model, state, noise * 80.0, goal, sig, persist=False)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), f"B={B}: max |d| = {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("fixture,n", [("g3_b256_lang.npz", 300), ("g1_tiny_mdtv.npz", 1000)])
def test_persistent_soak_is_bit_stable(fixture, n):
    """Stale-read detector: n back-to-back calls (no synchronisation in between) must all equal the first."""
    meta, fx = load_fixture(fixture)
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = torch.from_numpy(fx["sigmas"])
    x_T = noise * meta["sigma_max"]
    _need_kernel()
    _lib().mdt_op_set_persist(1)
    gs = sampling()
    with torch.no_grad():
        first = gs.sample_ddim(model, state, x_T, goal, sig).clone()
        bad = torch.zeros((), device="cuda", dtype=torch.int64)
        for _ in range(n):
            out = gs.sample_ddim(model, state, x_T, goal, sig)
            bad += (out != first).any().to(torch.int64)
    torch.cuda.synchronize()
    assert _engine(model).persist_status() == 0
    assert int(bad.item()) == 0, f"{int(bad.item())} of {n} calls differ from the first"
    assert_close(first.cpu(), fx["actions"], what="soak result vs reference")


@pytest.mark.parametrize("fixture", ["g3_b256_lang.npz", "g1_tiny_mdtv.npz", "g3_b8_smin1.npz"])
@pytest.mark.parametrize("persist", [True, False])
def test_device_sigmas_as_the_agent_passes_them(fixture, persist):
    """MDTVAgent.get_noise_schedule builds the schedule ON the device (mdtv_agent.py:660-667): it is consumed in place
    (mdt_sample_ddim_dev), no copy to the host; same actions as with a host schedule."""
    meta, fx = load_fixture(fixture)
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    sig = gs.get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"]) if "sigmas" not in fx else torch.from_numpy(fx["sigmas"])
    x_T = noise * meta["sigma_max"]
    if persist:
        _need_kernel()
    _lib().mdt_op_set_persist(1 if persist else 0)
    with torch.no_grad():
        host = gs.sample_ddim(model, state, x_T, goal, sig)
        dev = gs.sample_ddim(model, state, x_T, goal, sig.cuda())
    torch.cuda.synchronize()
    assert_close(dev.cpu(), fx["actions"], what="device schedule vs reference")
    assert_close(dev.cpu(), host.cpu(), rtol=1e-5, atol=1e-6, what="device vs host schedule")


def test_status_word_reports_nothing_on_a_healthy_run():
    meta, fx = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    eng = _engine(model)
    assert eng.persist_status() == 0


def test_graphed_sampler_replays_the_fused_call_bit_for_bit():
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    """GraphedDDIM: one fused sampler call captured as a HIP graph.  Replays with NEW inputs equal the eager call on those inputs
    bit for bit; a parameter update is seen (the re-upload runs outside the graph); growing the workspace re-captures."""
    from mdt_policy_amd.models.edm_diffusion.graphed import GraphedDDIM
    cfg = configs.mdtv_default()
    model = GCDenoiser(cfg, 0.5)
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}, strict=False)
    model = model.cuda().eval()
    sig = gs.get_sigmas_exponential(10, 0.001, 80.0).cuda()

    def inputs(seed, B=1):
        inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, seed).items()}
        return {"state_images": inp["state_images"], "modality": "lang"}, inp["noise"] * 80.0, inp["goal"]

    with torch.no_grad():
        st, x, goal = inputs(1)
        g = GraphedDDIM(model, st, x, goal, sig)
        for seed in (1, 2, 3):
            st, x, goal = inputs(seed)
            want = gs.sample_ddim(model, st, x, goal, sig).clone()
            got = g(st, x, goal, sig)
            assert torch.equal(got, want), f"seed {seed}"
            assert torch.equal(model.inner_model.latent_encoder_emb, model.inner_model.latent_encoder_emb)
        # parameter update (version counter bumps): the replay must follow
        model.inner_model.action_pred.weight.mul_(0.5)
        want = gs.sample_ddim(model, st, x, goal, sig).clone()
        assert not torch.equal(want, got)
        assert torch.equal(g(st, x, goal, sig), want)
        # a bigger batch elsewhere re-allocates the workspace: the graph is captured again on the next call
        st8, x8, goal8 = inputs(4, B=64)
        gs.sample_ddim(model, st8, x8, goal8, sig)
        want = gs.sample_ddim(model, st, x, goal, sig).clone()
        assert torch.equal(g(st, x, goal, sig), want)
        assert not g.matches(st8, x8, goal8, sig) and g.matches(st, x, goal, sig)


def test_rollout_sized_calls_switch_to_graph_replay_by_themselves(monkeypatch):
    """MDT_HIP_GRAPH unset ("auto"): the third call with the same rollout-sized shapes (B <= 8, eval, no grad) and every one
    after it replays a captured graph -- same bits as the eager launches; a large batch never does."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    cfg = configs.mdtv_default()
    model = GCDenoiser(cfg, 0.5)
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}, strict=False)
    model = model.cuda().eval()
    sig = gs.get_sigmas_exponential(10, 0.001, 80.0).cuda()

    def inputs(seed, B):
        inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, seed).items()}
        return {"state_images": inp["state_images"], "modality": "lang"}, inp["noise"] * 80.0, inp["goal"]

    monkeypatch.setattr(gs, "_GRAPH_MODE", "0")
    monkeypatch.setattr(gs, "_GRAPH_SAMPLER", False)
    eager = []
    with torch.no_grad():
        for seed in range(6):
            eager.append(gs.sample_ddim(model, *inputs(seed, 1), sig).clone())
    assert not model.__dict__.get("_graphed_samplers")
    monkeypatch.setattr(gs, "_GRAPH_MODE", "auto")
    with torch.no_grad():
        for seed in range(6):
            out = gs.sample_ddim(model, *inputs(seed, 1), sig)
            assert torch.equal(out, eager[seed]), seed
            assert bool(model.__dict__.get("_graphed_samplers")) == (seed >= gs._GRAPH_AUTO_AFTER), seed
        for _ in range(4):
            gs.sample_ddim(model, *inputs(9, 64), sig)
    assert len(model._graphed_samplers) == 1


def test_a_scaler_argument_keeps_the_fused_loop(monkeypatch):
    """The reference's DDIM accepts `scaler` and never reads it (gc_sampling.py:922-951), so a harness run with `use_scaler`
    (mdtv_agent.py:606-614) must get the same ONE native call -- not the per-step Python loop -- and the same bits."""
    meta, fx = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    sig = gs.get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    x_T = noise * meta["sigma_max"]
    fused, stepped = [], []
    native, fwd = model.sample_ddim, type(model).forward
    monkeypatch.setattr(model, "sample_ddim", lambda *a, **k: (fused.append(1), native(*a, **k))[1])
    monkeypatch.setattr(type(model), "forward", lambda self, *a, **k: (stepped.append(1), fwd(self, *a, **k))[1])

    class Scaler:  # any object: reading an attribute of it would raise
        def __getattr__(self, name):
            raise AssertionError(f"sample_ddim read scaler.{name}")

    with torch.no_grad():
        plain = gs.sample_ddim(model, state, x_T, goal, sig)
        scaled = gs.sample_ddim(model, state, x_T, goal, sig, scaler=Scaler())
    torch.cuda.synchronize()
    assert len(fused) == 2 and not stepped, f"native calls {len(fused)}, per-step denoiser calls {len(stepped)}"
    assert torch.equal(plain, scaled)
    assert_close(scaled.cpu(), fx["actions"], what="scaler passed vs reference")
    with torch.no_grad():  # a callback still needs the step loop, and gets every step
        seen = []
        gs.sample_ddim(model, state, x_T, goal, sig, scaler=Scaler(), callback=lambda d: seen.append(d["i"]))
    assert seen == list(range(meta["n_steps"])) and len(stepped) == meta["n_steps"]


@pytest.mark.parametrize("batch", [1, 3, 40])
def test_side_jobs_ride_in_neighbouring_launches_and_change_no_bit(batch):
    """Round 5: the products of a sampler call that do not depend on their neighbours in the launch chain (the sigma-MLP / adaLN
    table, M = n_steps rows; the MDTV token embedding) ride as extra workgroups in the next split-K small-M launch
    (mdt_gemm_side_*, k_gemm_smallm2).  Same tiles in the same order: the call's output is bit-identical with the switch off; at a
    rollout batch some launches must actually have taken one along, at a batch whose encoder runs on the tiled kernels the queue is
    simply launched behind it."""
    meta, fx = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    state = {k: (v[:batch] if torch.is_tensor(v) else v) for k, v in state.items()}
    goal, x_T = goal[:batch], noise[:batch] * meta["sigma_max"]
    gs = sampling()
    sig = gs.get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    lib = _lib()
    out = {}
    paired = {}
    with torch.no_grad():
        for on in (0, 1, 0, 1):
            lib.mdt_op_set_side_jobs(on)
            before = lib.mdt_op_side_jobs_paired()
            y = gs.sample_ddim(model, state, x_T, goal, sig)
            torch.cuda.synchronize()
            paired.setdefault(on, []).append(lib.mdt_op_side_jobs_paired() - before)
            out.setdefault(on, []).append(y.clone())
    assert paired[0] == [0, 0], paired
    if batch <= 3:
        assert paired[1][0] >= 3 and paired[1][0] == paired[1][1], paired   # the three table products at least
    for a in out[0] + out[1]:
        assert torch.equal(a, out[0][0]), "side jobs changed the result"
    assert_close(out[1][0].cpu(), fx["actions"][:batch], what="side jobs on vs reference")
