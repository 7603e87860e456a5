"""Summarise rocprofv3 --pmc CSVs of tools/gpu_pmc.sh into one table per kernel (averages over dispatches)."""
import collections, csv, glob, os, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "k_gemm" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]) // int(r["Workgroup_Size"]), r["LDS_Block_Size"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(f"## {k[0]}  workgroups={k[1]} lds={k[2]}")
    for c in sorted(m): print(f"   {c:34s} {m[c]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print(f"   -> MFMA pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles per SIMD (= 32 x MFMA instructions per SIMD); "
              "divide by kernel time x shader clock for the pipe utilisation")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        print("   -> wave-cycle split: " + ", ".join(f"{n} {m.get(n, 0) / wc:.2f}" for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")))
    if "TCC_EA0_RDREQ_sum" in m:
        print(f"   -> L2: hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}; EA read requests {m['TCC_EA0_RDREQ_sum']:.0f} (x128 B = {m['TCC_EA0_RDREQ_sum'] * 128 / 1e6:.1f} MB), write requests {m['TCC_EA0_WRREQ_sum']:.0f} (x64 B = {m['TCC_EA0_WRREQ_sum'] * 64 / 1e6:.1f} MB)  [units calibrated on the N=384 projection: writes == 3.93 MB output]")
