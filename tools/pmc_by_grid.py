"""Mean of one rocprofv3 --pmc counter per (kernel, grid) from ``*counter_collection.csv`` files.  ``python tools/pmc_by_grid.py <dir> [<dir> ...]``"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("qv2x::", "").replace("(anonymous namespace)::", "").replace("void ", "")
            grid = r.get("Grid_Size") or "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            agg[(r["Counter_Name"], name.split("(")[0][:60], grid)].append(float(r["Counter_Value"]))
    print("counter;kernel;grid;launches;mean")
    for (c, k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= 3:
            print(f"{c};{k};{g};{len(v)};{sum(v) / len(v):.1f}")
