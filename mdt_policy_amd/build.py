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
SOURCES = ["mdt_kernels.hip", "mdt_model.hip", "mdt_resampler.hip", "mdt_map_pool.hip", "mdt_mae.hip", "mdt_infonce.hip", "mdt_train_kernels.hip", "mdt_train_ops.hip", "mdt_train.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 kernels)")
    return exe


HEADERS = ["mdt_internal.h", "mdt_device.h", "mdt_model_types.h", "mdt_tiles.h", "mdt_tall.h", "mdt_ws.h", "mdt_mlp_split.h"]
PUBLIC_HEADERS = ["mdt_hip.h", "mdt_hip_ops.h", "mdt_resampler.h", "mdt_map_pool.h", "mdt_hip_train.h", "mdt_mae.h"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(INCLUDE, f) for f in PUBLIC_HEADERS]
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(args):
    src, obj, defines, verbose = args
    tmp = f"{obj}.tmp.{os.getpid()}"  # compile beside the target, then rename: several ranks that find a stale library and
    #                                   build at the same time never link each other's half-written objects
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", "-Wall", "-Wno-unused-function",
           f"-I{INCLUDE}", f"-I{CSRC}", "-o", tmp] + [f"-D{d}" for d in defines] + [src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + res.stdout + res.stderr)
    os.replace(tmp, obj)
    return res.stderr


def build_library(force: bool = False, verbose: bool = False, out: str = None, defines=()) -> str:
    """Compile csrc/*.hip -> csrc/libmdt_hip.so for gfx950; returns the library path.  Every translation unit is compiled
    to its own object (in parallel; objects are kept under csrc/build/ and reused while the source and the headers are
    older), then linked.  ``out`` / ``defines`` build an experimental variant next to the product library (tuning A/B runs)."""
    if out is None and not force and not needs_build():
        return LIB
    out = out or LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    tag = hashlib.sha1(" ".join(sorted(defines)).encode()).hexdigest()[:8] if defines else "default"
    objdir = os.path.join(CSRC, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(p) for p in [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, h) for h in PUBLIC_HEADERS]
                + [os.path.abspath(__file__)])
    jobs, objs = [], []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            jobs.append((src, obj, tuple(defines), verbose))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        for warn in ex.map(_compile_one, jobs):
            if verbose and warn.strip():
                print(warn, file=sys.stderr)
    tmp = f"{out}.tmp.{os.getpid()}"  # link next to the target, then rename: concurrent importers never see a torn file
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build_library(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None, defines=defs))
