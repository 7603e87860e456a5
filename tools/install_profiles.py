#!/usr/bin/env python3
"""Copy the summaries of one tools/gpu_final.sh run (gpurun_out/<tag>/) into profiles/r<NN>_*, each with a header naming the
round, the commit the run was built from and the command.   usage: python tools/install_profiles.py <tag> <round> <commit>"""
import os
import sys

tag, rnd, commit = sys.argv[1], int(sys.argv[2]), sys.argv[3]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", tag)
dst = os.path.join(R, "profiles")
pre = f"r{rnd:02d}_"
head = f"# Round {rnd} (final, commit {commit}) -- "


def read(name):
    with open(os.path.join(src, name)) as f:
        return f.read()


def lines(name, pat):
    return "".join(l for l in read(name).splitlines(True) if pat in l)


def write(name, text):
    with open(os.path.join(dst, pre + name), "w") as f:
        f.write(text)
    print("wrote", pre + name)


bench = read("bench.json").strip().splitlines()[-1]
write("bench.json", bench + "\n")
stats = read("bench_kernel_stats.txt").splitlines(True)
# sampler calls of the profiled command = launches of the once-per-call fold kernel (2 warm-up + 5 timed + 50 spread + 2 of the
# in-chain dominant-kernel leg)
calls = next((l.split("|")[1].strip() for l in stats if "k_xattn_fold" in l), "?")
write("bench_kernel_stats.txt",
      head + "rocprofv3 --kernel-trace --stats of 'python bench.py --steps 5 --warmup 2 --no-cpu-baseline' (MI355X, 1 GPU), "
      f"tools/gpu_final.sh\n# {calls} sampler calls x (encoder + fold + 10 decoder steps) (warm-up, timed, the 50-call spread leg, the "
      "traced call) + the 210-launch dominant-kernel-alone leg; bench "
      f"line of the same build and box: profiles/{pre}bench.json\n" + "".join(stats[:32]))
write("bench_pmc.txt",
      head + "rocprofv3 --kernel-trace --pmc (two separate passes) of 'python bench.py --steps 3 --warmup 1 --no-cpu-baseline', "
      "tools/gpu_final.sh\n" + read("bench_pmc.txt"))
lat = read("lat_kernel_stats.txt").splitlines(True)
write("rollout_b1_kernel_stats.txt",
      head + "rollout batch B = 1: rocprofv3 --kernel-trace --stats of 'python tools/latency.py 1' (sampler calls under the "
      "profiler), MI355X\n" + lines("lat_run.txt", "B=") + "# per-kernel summary of the whole run\n" + "".join(lat[:17]) +
      "# the LAST call (250 dispatches): durations and the idle gap in front of each kernel (tools/prof_gaps.py)\n" +
      "".join(read("lat_gaps.txt").splitlines(True)[:16]))
if os.path.exists(os.path.join(src, "lat_call.txt")):
    write("rollout_b1_call.txt", head + "rollout batch B = 1: the LAST sampler call of 'python tools/latency.py 1' under rocprofv3 --kernel-trace, launch "
          "by launch (tools/prof_call.py): start, duration, idle gap in front, workgroups x threads, kernel\n" + read("lat_call.txt"))
write("batch_sweep.txt",
      f"# tools/latency.py on MI355X (end of round {rnd}, commit {commit}): one sample_ddim call = encoder + 10 DDIM steps, MDT-V "
      "default, fp32\n" + read("sweep.txt"))
tr = read("train_kernel_stats.txt").splitlines(True)
write("train_step_kernel_stats.txt",
      head + "rocprofv3 --kernel-trace --stats of 'MDT_TRAIN_BENCH_MODES=train MDT_TRAIN_BENCH_OPT=fused python tools/train_bench.py 1024' (loss "
      "forward + backward + FusedAdamW at B = 1024, train mode; times under the profiler; the weight gradients run on a side stream: per-kernel durations overlap), MI355X\n" + lines("train_run.txt", "B=") +
      "# without the profiler (same box):\n" + "".join("# " + l for l in read("train_bench.txt").splitlines(True)) + "".join(tr[:36]))
mae = read("mae_kernel_stats.txt").splitlines(True)
write("mae_kernel_stats.txt",
      head + "masked generative foresight head: rocprofv3 --kernel-trace --stats of 'python tools/mae_bench.py 1024' (forward + "
      "backward, B = 1024), MI355X\n# without the profiler (same box):\n" + read("mae_bench.txt") + "".join(mae[:34]))
# PMC passes of the training step and of the masked-image head (tools/gpu_train_pmc.sh: MFMA-busy | HBM-side requests + L2 hit)
for name, what in (("train", "MDT_TRAIN_BENCH_MODES=train MDT_TRAIN_BENCH_OPT=fused python tools/train_bench.py 1024' (denoiser training step, B = 1024, train mode, FusedAdamW; weight gradients on the side stream"),
                   ("mae", "python tools/mae_bench.py 1024' (masked-image head, forward + backward, B = 1024: 104448 decoder rows")):
    fn = os.path.join("trainpmc", f"{name}_pmc.txt")
    if os.path.exists(os.path.join(src, fn)):
        write(f"{'train_step' if name == 'train' else 'mae'}_pmc.txt",
              head + f"rocprofv3 --pmc over '{what}), two separate passes with kernel-trace only (tools/gpu_train_pmc.sh):\n"
              "# SQ_VALU_MFMA_BUSY_CYCLES | TCC_EA0_RDREQ / WRREQ + TCC_HIT / MISS; HBM-side MB = RDREQ x 128 B / WRREQ x 64 B per launch "
              "(MI355X_MICROARCH.md, gfx950 corrections).  Reading: DESIGN.md section 5a.\n" + read(fn))
# the dominant kernel's HBM-side traffic per launch, for bench.py's roofline.traffic (PMC counters cannot be read from inside
# the bench process): the fused MLP launch's row of the PMC table above
import json
import re
whole = re.search(r"whole run: MFMA busy ([0-9.]+)", read("bench_pmc.txt"))
for l in read("bench_pmc.txt").splitlines():
    if l.startswith("k_mlp<") or l.startswith("k_mlp_split<"):
        f = [x.strip() for x in l.split("|")]
        rd, wr = float(f[6]), float(f[7])
        M, D, N, S = 2560, 384, 1536, 3
        split = l.startswith("k_mlp_split<")
        wb = 6 if split else 4   # bytes per weight of the image the kernel reads (split: three bf16 parts)
        write("dominant_kernel_pmc.json", json.dumps({
            "commit": commit,
            "kernel": f"{f[0]} (LN + modulate -> c_fc -> GELU -> c_proj -> gate, {M} rows, d = {D}, hidden {N}; B = 256)",
            "source": f"rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum of the bench command "
                      f"itself (own pass, tools/gpu_final.sh; table profiles/{pre}bench_pmc.txt), per launch, {f[2]} dispatches",
            "read_bytes": int(rd * 1e6), "write_bytes": int(wr * 1e6), "hbm_side_bytes_per_launch": int((rd + wr) * 1e6),
            "l2_hit": float(f[8]), "mfma_busy_frac_at_2.4GHz": float(f[5]), "avg_us_under_pmc": float(f[3]),
            "whole_run_mfma_busy": float(whole.group(1)) if whole else None,
            "units": "guide MI355X_MICROARCH.md 'HBM': EA read requests are 128-B requests on gfx950 (FETCH_SIZE = RDREQ x 64 B reports "
                     "half) -> x128 B; write requests x64 B (calibrated in round 1 on a GEMM whose output size is exact)",
            "algorithmic_bytes_per_launch": 4 * (M * D + M * D) + wb * 2 * N * D,
            "algorithmic_bytes_per_launch_with_slabs": 4 * (M * D + S * M * D) + wb * 2 * N * D,
            "note": ("split form (round 6): the two weight images are 3.5 MB each (6 bytes per weight), read by slice: the workgroups of a hidden slice "
                     "(2.36 MB of the two images) run on neighbouring XCDs, so an L2 fetches one or two slices, not all three; every slice's "
                     "workgroups read the 3.9 MB of rows again; the writes are exactly the three partial slabs (3 x 3.9 MB).  The hidden layer "
                     "never leaves the CU.") if split else "reads exceed the truly algorithmic 12.6 MB (3.9 MB of rows + 4.7 MB of weights read once, 3.9 MB written) because each of the 8 XCD L2s fetches both 2.36 MB weight images once (served "
                    "by the 256 MB Infinity Cache, which these memory-side counters include): 8 x 4.7 + 3.9 MB of rows; the writes are "
                    "exactly the three partial slabs (3 x 3.9 MB).  The hidden layer (15.7 MB per launch in round 2) never leaves the CU."
        }, indent=2) + "\n")
        break
