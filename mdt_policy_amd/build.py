"""Build libmdt_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libmdt_hip.so")
SOURCES = ["mdt_kernels.hip", "mdt_persist.hip", "mdt_model.hip", "mdt_resampler.hip", "mdt_map_pool.hip", "mdt_mae.hip", "mdt_infonce.hip", "mdt_train_kernels.hip", "mdt_train_ops.hip", "mdt_train.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 kernels)")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + ["mdt_internal.h", "mdt_device.h", "mdt_model_types.h", "mdt_tiles.h", "mdt_persist.h"]]
    deps += [os.path.join(INCLUDE, f) for f in ("mdt_hip.h", "mdt_hip_ops.h", "mdt_resampler.h", "mdt_map_pool.h", "mdt_hip_train.h", "mdt_mae.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, out: str = None, defines=()) -> str:
    """Compile csrc/*.hip -> csrc/libmdt_hip.so for gfx950; returns the library path.
    ``out`` / ``defines`` build an experimental variant next to the product library (tuning A/B runs)."""
    if out is None and not force and not needs_build():
        return LIB
    out = out or LIB
    tmp = f"{out}.tmp.{os.getpid()}"  # compile next to the target, then rename: concurrent importers never see a torn file
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}", "-o", tmp] + [f"-D{d}" for d in defines]
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, out)
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build_library(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None, defines=defs))
