#!/bin/bash
# round 6: SQ counters of the split weight-stationary body (where its issue slots go)
OUT=$PWD/gpurun_out/splitpmc; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $OUT/p1 -o p -- python $R/tools/ws_split_bench.py > $OUT/p1.txt 2> $OUT/p1.err
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "/root/repo/gpurun_out/splitpmc")
f = glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = (r["Kernel_Name"][:60], r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in agg.items():
    if "ws" not in k[0]: continue
    print(k, "launches", n[k])
    for name, v in sorted(c.items()): print(f"   {name:28s} {v / max(n[k], 1):14.0f}")
PY
