"""Per-kernel averages of every counter in the counter_collection CSVs under a directory (rocprofv3 --pmc passes), one row per
(kernel, workgroups), sorted by total duration.  usage: python tools/pmc_table.py <dir> [rows=25]"""
import collections, csv, glob, os, sys
root = sys.argv[1]; nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
val = collections.defaultdict(lambda: collections.defaultdict(float)); num = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float); nd = collections.defaultdict(int); seen = set()
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        k = (name[:48], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        c = r["Counter_Name"]
        val[k][c] += float(r["Counter_Value"]); num[k][c] += 1
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; nd[k] += 1
ctrs = sorted({c for d in val.values() for c in d})
print("kernel | workgroups | dispatches | avg_us | " + " | ".join(ctrs))
for k in sorted(val, key=lambda k: -dur[k])[:nrows]:
    print(f"{k[0]} | {k[1]} | {nd[k]} | {dur[k] / max(1, nd[k]):.1f} | " + " | ".join(f"{val[k][c] / max(1, num[k][c]):.4g}" for c in ctrs))
