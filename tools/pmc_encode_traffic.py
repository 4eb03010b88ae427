"""HBM-side traffic of the dominant kernel (codebook_encode_wave_kernel at the bench's batch) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the
bench command -> profiles/rNN_pmc_encode.json (the figure bench.py's roofline.traffic quotes).  FETCH_SIZE / WRITE_SIZE are in KiB.
Only the launches of the bench's own batch (the largest grid) are averaged.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half
the bytes of wide coalesced streaming reads -- the corrected figure doubles it; WRITE_SIZE is taken as reported."""
import csv, glob, json, sys
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
def mean(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "codebook_encode" in r["Kernel_Name"] and "_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                if "codebook_encode_wave_kernel" in r["Kernel_Name"]:        # the many-frames form: one 64-thread wave per 32 rows
                    nrows = int(r["Grid_Size"]) // 64 * 32
                else:                                                       # rows per 512-thread workgroup (the kernel's template argument)
                    nrows = int(r["Grid_Size"]) // 512 * (64 if "<64>" in r["Kernel_Name"] else 32)
                rows.append((nrows, float(r["Counter_Value"]), int(r["Grid_Size"])))
    if not rows:
        return None, 0, 0, 0
    g = max(r[0] for r in rows)                                            # the launches with the most rows: the bench's own batch
    v = [c for (nr, c, gs) in rows if nr == g]
    return sum(v) / len(v), len(v), [gs for (nr, c, gs) in rows if nr == g][0], g
f, nf, gf, rows_f = mean(fetch_dir, "FETCH_SIZE")
w, nw, gw, rows_w = mean(write_dir, "WRITE_SIZE")
res = {"kernel": "codebook_encode_wave_kernel: the whole rounds of waves of the bench's batch (34 816 of its 35 200 waves' worth of rows; the remainder runs as 384 workgroups of codebook_encode_kernel<32>, not counted here)", "launches_averaged": [nf, nw], "grid_threads": gf, "rows_per_launch": rows_f, "agent_frames_per_launch": round(rows_f / 35200, 3) if gf else None,
       "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
       "traffic_bytes_per_launch_raw": None if f is None or w is None else int((f + w) * 1024),
       "traffic_bytes_per_launch": None if f is None or w is None else int((2 * f + w) * 1024),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras` (tools/final_evidence.sh); "
               "the launches of the bench's own batch only; traffic_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB (gfx950: FETCH_SIZE counts "
               "64 B per 128-B request, MI355X_MICROARCH.md), _raw = the counters as reported"}
json.dump(res, open(out, "w"), indent=1)
print(res)
