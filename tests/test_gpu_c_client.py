"""The C ABI driven by a plain C host program (tests/c_client/client.c, no Python / torch in that process) gives the
same actions as the Python facade on the same weights and inputs -- the boundary another host language would bind."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from mdt_policy_amd import _lib, build
from tests.helpers import cfg_of, inputs_of, load_fixture, params_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fixture", ["g1_tiny_mdtv.npz", "g1_tiny_mdt.npz"])
def test_c_host_program_matches_the_python_facade(fixture, tmp_path):
    exe = tmp_path / "client"
    lib = _lib.library_path()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run([shutil.which("gcc") or "gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(ROOT, "tests", "c_client", "client.c"), "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(rocm, "include"), "-o", str(exe), lib, "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.join(rocm, "lib")], check=True)
    meta, fx = load_fixture(fixture)
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    model = GCDenoiser(cfg_of(meta), 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    cfg = model.inner_model._hip_config(0.5)
    state, goal, noise = inputs_of(meta)
    sig = gs.get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    x_T = noise * meta["sigma_max"]
    blob = tmp_path / "blob.bin"
    allf = [n for n, _ in _lib.MDTConfig._fields_]
    names = allf[:allf.index("sigma_data")]  # the int32 fields that lead the struct
    with open(blob, "wb") as f:
        f.write(struct.pack("<i", len(names)))
        f.write(struct.pack(f"<{len(names)}i", *[getattr(cfg, n) for n in names]))
        f.write(struct.pack("<f", 0.5))
        sd = {"inner_model." + k: v for k, v in model.inner_model.state_dict().items()}
        eng = model.inner_model.hip_engine(0.5)
        wanted = [k for k in eng.expected]
        f.write(struct.pack("<i", len(wanted)))
        for k in wanted:
            t = sd[k].detach().cpu().float().contiguous().numpy()
            f.write(struct.pack("<i", len(k)) + k.encode() + struct.pack("<q", t.size) + t.tobytes())
        B = x_T.shape[0]
        f.write(struct.pack("<ii", B, meta["n_steps"]) + sig.numpy().astype(np.float32).tobytes())
        if meta["arch"] == "mdtv":
            f.write(state["state_images"].numpy().tobytes())
        else:
            f.write(state["static"].numpy().tobytes() + state["gripper"].numpy().tobytes())
        f.write(goal.numpy().tobytes() + x_T.numpy().tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(blob), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "gfx950" in r.stdout
    got = np.fromfile(out, dtype=np.float32)
    nact = x_T.numel()
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    with torch.no_grad():
        want = gs.sample_ddim(model, gstate, x_T.cuda(), goal.cuda(), sig).cpu().numpy()
    np.testing.assert_array_equal(got[:nact].reshape(want.shape), want)          # same library, same kernels: bit exact
    np.testing.assert_allclose(got[:nact].reshape(want.shape), fx["actions"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(got[nact:].reshape(fx["ctx"].shape), fx["ctx"], rtol=1e-3, atol=1e-4)
