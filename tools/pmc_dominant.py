"""HBM-side traffic of the kernels that can be the bench's DOMINANT one (bench.py: rooflines -> `roofline`), from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE: separate runs of the bench command) -> profiles/rNN_pmc_dominant.json, keyed as bench.py looks them up:
  deconv            deconv_ps_batch_kernel             units = agent-frames of the launch
  encode_listed     codebook_encode_wave_kernel<.., true>   units = listed cells (read from the bench line of the same command)
  encode_candidates encode_candidates_kernel           units = agent-frames
  encode            codebook_encode_wave_kernel (every cell)  units = agent-frames
  conv_<layer>      conv3x3_i8_wide_kernel at the bench's batch: the two shrinker layers share a kernel and a grid -- the launches are split by their
                    FETCH_SIZE (the 384-channel input of double_conv.0 reads 1.5x the 256-channel one's)
FETCH_SIZE / WRITE_SIZE are KiB; traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB (gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md).
Only the launches with the kernel's largest grid (the bench's own batch) are averaged.
    python tools/pmc_dominant.py <fetch_dir> <write_dir> <bench_line.json> <out.json> [frames_per_launch=32]"""
import collections, csv, glob, json, sys
fetch_dir, write_dir, line_file, out = sys.argv[1:5]
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 32


def rows(d, counter):
    r_ = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                r_[r["Kernel_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return r_


def mean_largest(v):
    g = max(x[0] for x in v)
    vals = [x[1] for x in v if x[0] == g]
    # a persistent kernel has one grid whatever its work (the listed cells are a device-side count): of the largest grid's launches, only
    # those that moved more than half of what the busiest one moved -- the bench's own batch, not the one-frame launches of the latency loop
    vals = [x for x in vals if x > max(vals) / 2]
    return sum(vals) / len(vals), len(vals), g, vals


F, W = rows(fetch_dir, "FETCH_SIZE"), rows(write_dir, "WRITE_SIZE")
try:
    line = json.loads(open(line_file).read().strip().splitlines()[-1])
    refined = line["roofline_stages"]["codebook_encode_two_stage"]["refined_cells"]
except Exception:
    line, refined = None, None
res = {}


def put(key, match, units, unit, split=None):
    names = [k for k in F if match(k)]
    if not names:
        return
    fv = [x for k in names for x in F[k]]
    wv = [x for k in names for x in W.get(k, [])]
    if not fv or not wv:
        return
    f, nf, g, fvals = mean_largest(fv)
    w, nw, _, wvals = mean_largest(wv)
    if split is not None:                                 # the launches of one kernel / grid that belong to different layers: by FETCH_SIZE
        cut = (min(fvals) + max(fvals)) / 2
        sel = [v for v in fvals if (v > cut) == (split == "high")]
        f = sum(sel) / len(sel)
        nf = len(sel)
    res[key] = {"kernel": names[0][:120], "grid": g, "launches_averaged": nf, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                "traffic_bytes_per_launch": int((2 * f + w) * 1024), "units_per_launch": units, "unit": unit,
                "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB, two separate --pmc passes of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras`"}


put("deconv", lambda k: "deconv_ps_batch_kernel" in k, frames, "agent-frames")
put("encode_candidates", lambda k: "encode_candidates_kernel" in k, frames, "agent-frames")
if refined:
    put("encode_listed", lambda k: "codebook_encode_wave_kernel" in k and "true>" in k.replace(" ", ""), refined, "listed cells")
put("encode", lambda k: "codebook_encode_wave_kernel" in k and "true>" not in k.replace(" ", ""), frames, "agent-frames")
put("conv_shrinker_m1.layers.0.double_conv.0", lambda k: "conv3x3_i8_wide_kernel<true" in k.replace(" ", ""), frames, "agent-frames", split="high")
put("conv_shrinker_m1.layers.0.double_conv.1", lambda k: "conv3x3_i8_wide_kernel<true" in k.replace(" ", ""), frames, "agent-frames", split="low")
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    print(k, v["traffic_bytes_per_launch"], v["launches_averaged"], v["kernel"][:70])
