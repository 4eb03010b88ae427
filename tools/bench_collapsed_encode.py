"""The OPT-IN collapsed codebook encode (qv2x_codebook_encode_collapsed_f32) against the exact one (qv2x_codebook_encode_f32, the shipped
default) on the same int8 shrinker output: index mismatches, the exact path's top-2 distance gap on every mismatched cell (from the CPU
checker), kernel times, and the frame rate with either mode.

    python tools/bench_collapsed_encode.py [frames_for_the_mismatch_count]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    import bench
    from oracle.spec import Oracle
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = "cuda:0"
    state, eng, fp_model, qt = bench.build_engine(32)
    out = {"workload": "V2X-Real single-agent frames, 35 200 cells x 3 levels x 128 codewords"}
    orc = Oracle(state)
    mism_total, cells_total, gaps_mism, gaps_all = np.zeros(3, np.int64), 0, [], []
    full = bench.frame_batch(1, 0, frames, dev)[1]
    eng.encode_mode = "exact"
    exact = eng.encode_agents(full["inputs_m1"], frames).clone()
    eng.encode_mode = "collapsed"
    coll = eng.encode_agents(full["inputs_m1"], frames).clone()
    torch.cuda.synchronize()
    b = eng._workspace(frames)
    s1 = (b["s1"][:, 1:-1, 1:-1, :].to(torch.int16) + 128).cpu().numpy().astype(np.float32)
    q = eng.shrink1.out_q
    rows = ((s1 - np.float32(q[1])) * np.float32(q[0])).reshape(-1, 256)
    want, gaps = orc.encode_rows(rows, want_gaps=True)
    e, c = exact.cpu().numpy().reshape(3, -1), coll.cpu().numpy().reshape(3, -1)
    assert np.array_equal(e, want), "the exact kernel no longer matches the checker"
    # a mismatch at level l changes the residual every later level sees: only the FIRST differing level of a cell is a rounding event
    mm = e != c
    seen = np.zeros(mm.shape[1], bool)
    for l in range(3):
        first = mm[l] & ~seen
        seen |= mm[l]
        mism_total[l] = first.sum()
        gaps_mism.append(gaps[l][first])
    cells_total = e.shape[1]
    gaps_all.append(gaps.reshape(-1))
    # which of the two is the argmin in float64?  (the heads walked in the reference's order on the mismatched cells only)
    g64 = lambda l, n: state[f"codebook/{l}/{n}"].astype(np.float64)
    agree64 = {"collapsed": 0, "exact": 0, "neither": 0}
    rel_gap = []
    seen = np.zeros(mm.shape[1], bool)
    for l in range(3):
        first = np.nonzero(mm[l] & ~seen)[0]
        seen |= mm[l]
        for cell in first:
            x = rows[cell].astype(np.float64)
            for j in range(l + 1):
                z = g64(j, "stage_w") @ x + g64(j, "stage_b")
                qv = g64(j, "qhead_w") @ z + g64(j, "qhead_b")
                if j < l:
                    x = g64(j, "lhead_w") @ z + g64(j, "lhead_b") - g64(j, "codebook")[e[j, cell]]
            dist = ((qv[None, :] - g64(l, "codebook")) ** 2).sum(1)
            k64 = int(dist.argmin())
            agree64["collapsed" if k64 == c[l, cell] else "exact" if k64 == e[l, cell] else "neither"] += 1
            two = np.sort(dist)[:2]
            rel_gap.append(float((two[1] - two[0]) / two[0]))
    out["float64_argmin_agrees_with"] = agree64
    out["float64_relative_gap_of_mismatched_cells_max"] = max(rel_gap) if rel_gap else 0.0
    gm = np.concatenate(gaps_mism) if any(len(g) for g in gaps_mism) else np.zeros(0, np.float32)
    ga = np.concatenate(gaps_all)
    out["cells"] = int(cells_total)
    out["first_mismatch_per_level"] = [int(v) for v in mism_total]
    out["cells_with_any_mismatch_rate"] = float(mism_total.sum() / max(cells_total, 1))
    out["top2_gap_of_mismatched_cells_max"] = float(gm.max()) if gm.size else 0.0
    out["top2_gap_all_cells_median"] = float(np.median(ga))
    out["top2_gap_all_cells_p01"] = float(np.percentile(ga, 1))

    for mode in ("exact", "collapsed"):
        eng.encode_mode = mode
        for nb in (1, 8):
            full = bench.frame_batch(1, 0, nb, dev)[1]
            eng.encode_agents(full["inputs_m1"], nb)
            us = bench.event_time_us(lambda: eng.encode_codes(nb), 20)
            out[f"encode_us_per_frame_{mode}_batch{nb}"] = round(us / nb, 1)
        full = bench.frame_batch(1, 0, 8, dev)[1]
        rep = eng.capture(full)
        us = bench.event_time_us(rep, 20)
        out[f"frame_us_{mode}_batch8_one_stream"] = round(us / 8, 1)
    eng.encode_mode = "exact"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
