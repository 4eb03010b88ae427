import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "qv2x" not in k: continue
            print(k)
            for c, v in sorted(cs.items()): print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
