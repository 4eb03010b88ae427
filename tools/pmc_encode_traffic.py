"""HBM-side traffic of the dominant kernel (codebook_encode_kernel) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the
bench command -> profiles/r02_pmc_encode.json (the figure bench.py's roofline.traffic quotes).  FETCH_SIZE / WRITE_SIZE are in KiB."""
import csv, glob, json, sys
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
def mean(d, counter):
    v = []
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "codebook_encode_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) > 512 * 2000:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)
f, nf = mean(fetch_dir, "FETCH_SIZE")
w, nw = mean(write_dir, "WRITE_SIZE")
res = {"kernel": "codebook_encode_kernel", "launches_averaged": [nf, nw], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
       "traffic_bytes_per_launch": None if f is None or w is None else int((f + w) * 1024),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline`; "
               "batch-of-8 launches only (grid > 1M threads); raw counter values x 1 KiB, no gfx950 doubling applied (see DESIGN.md §5)"}
json.dump(res, open(out, "w"), indent=1)
print(res)
