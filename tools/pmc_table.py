"""rocprofv3 --pmc output -> one row per (kernel, grid) with every counter as a column (means over launches).
    python tools/pmc_table.py <dir> [min_launches]"""
import collections, csv, glob, sys
d = sys.argv[1]; minl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("qv2x::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        grid = r.get("Grid_Size") or "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        agg[(name, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = sorted({c for v in agg.values() for c in v})
print(";".join(["kernel", "grid", "launches"] + cols))
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values())):
    n = max(len(x) for x in v.values())
    if n >= minl:
        print(";".join([k, g, str(n)] + [f"{sum(v[c]) / len(v[c]):.0f}" if c in v else "" for c in cols]))
