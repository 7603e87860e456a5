"""Per-kernel and whole-run MFMA-busy / HBM-side traffic from the two --pmc passes of tools/gpu_bench_pmc.sh.
MFMA-busy fraction = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (kernel duration x 2.4 GHz): the share of the peak
MFMA issue slots (the clock behind the 157.3 TFLOP/s figure) the kernel used; durations from the dispatch timestamps
of the same (profiled) run.  GRBM_GUI_ACTIVE is not used: it is not a per-dispatch cycle count in this rocprofv3."""
import collections, csv, glob, os, sys
root = sys.argv[1]
WHAT = sys.argv[2] if len(sys.argv) > 2 else "'bench.py --steps 3 --warmup 1' (4 sampler calls + the 210-launch dominant-kernel leg)"
NROWS = int(sys.argv[3]) if len(sys.argv) > 3 else 14
CLK = 2.4e9
val = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> counter -> sum
num = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float)                                   # kernel -> summed duration (s), pass 1 only
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if name.startswith("at::") or "rocclr" in name or name.startswith("k_pack") or "elementwise" in name:
            name = "(other)"
        k = (name, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))  # rocprofv3's Grid_Size / Workgroup_Size are the products over x, y, z
        c = r["Counter_Name"]
        val[k][c] += float(r["Counter_Value"]); num[k][c] += 1
        if c == "SQ_VALU_MFMA_BUSY_CYCLES":
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
rows = sorted(val.items(), key=lambda kv: -dur[kv[0]])
print(f"# sums over all dispatches of {WHAT}, profiled run")
print("kernel | workgroups | dispatches | avg us | MFMA-busy cycles/SIMD/launch | MFMA-busy frac @2.4GHz | HBM-side read MB/launch | write MB/launch | L2 hit | HBM-side TB/s | share of kernel time")
tb = td = 0.0
for k, d in rows[:NROWS]:
    n = max(1, num[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024
    frac = busy / (dur[k] * CLK) if dur[k] else 0.0
    tb += busy; td += dur[k]
    rd = d.get("TCC_EA0_RDREQ_sum", 0.0) * 128 / 1e6 / max(1, num[k].get("TCC_EA0_RDREQ_sum", 1))
    wr = d.get("TCC_EA0_WRREQ_sum", 0.0) * 64 / 1e6 / max(1, num[k].get("TCC_EA0_WRREQ_sum", 1))
    hit = d.get("TCC_HIT_sum", 0.0) / max(1.0, d.get("TCC_HIT_sum", 0.0) + d.get("TCC_MISS_sum", 0.0))
    tbs = (rd + wr) * 1e6 / (dur[k] / n) / 1e12 if dur[k] else 0.0
    print(f"{k[0][:44]} | {k[1]} | {n} | {dur[k] / n * 1e6:.2f} | {busy / n:.0f} | {frac:.3f} | {rd:.1f} | {wr:.1f} | {hit:.2f} | {tbs:.2f} | {100 * dur[k] / max(1e-12, sum(dur.values())):.1f}%")
allb = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in val.values()) / 1024
alld = sum(dur.values())
rd = sum(d.get("TCC_EA0_RDREQ_sum", 0.0) for d in val.values()) * 128 / 1e9
wr = sum(d.get("TCC_EA0_WRREQ_sum", 0.0) for d in val.values()) * 64 / 1e9
print(f"# whole run: MFMA busy {allb / (alld * CLK):.3f} of the peak issue slots over {alld * 1e3:.1f} ms of kernel time; "
      f"HBM-side traffic {rd:.2f} GB read + {wr:.2f} GB written = {(rd + wr) / alld / 1e3:.2f} TB/s of 8 TB/s")
