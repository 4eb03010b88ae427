"""Time the quantized SECOND encoder at full size on one MI355X: eager launches and one hipGraph replay.

    python tools/bench_second.py [n_points] [--check]      (--check: also compare with oracle/spec_second.py at this size)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch


def main():
    from _common_second import calibrated_second, second_scene_np
    from quantv2x_amd import synth
    from quantv2x_amd.engine_second import DeployedSecondEncoder
    from quantv2x_amd.ptq_state import export_second_state
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60000
    shape = "second_full"
    t0 = time.time()
    qm = calibrated_second(shape, 1, n_points)
    state = export_second_state(qm.model.encoder_m1)
    sc = second_scene_np(shape, 1, n_points)
    cal_s = time.time() - t0
    eng = DeployedSecondEncoder(state, "cuda:0", agents=1, max_voxels=synth.SECOND_SHAPES[shape][2])
    inp = {k: torch.from_numpy(v).cuda() for k, v in sc.items()}
    taps = {}
    eng(inp, taps)
    torch.cuda.synchronize()
    sites = [int(taps[f"second/{i}"][2].item()) for i in range(12)]
    macs = 0
    for i, ly in enumerate(eng.layers):
        macs += sites[i] * ly.K * ly.ci * ly.co
    out = {"voxels": int(sc["voxel_coords"].shape[0]), "sites_per_layer": sites, "dense_window_gmac": macs / 1e9, "calibration_s": round(cal_s, 1)}
    if "--check" in sys.argv:
        from oracle.spec_second import OracleSecond
        t0 = time.time()
        want = OracleSecond(state).forward(sc, batch_size=1)
        out["oracle_s"] = round(time.time() - t0, 1)
        out["bit_exact"] = bool(np.array_equal(eng.dense_codes().cpu().numpy(), want))

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    out["eager_ms"] = timed(lambda: eng(inp))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng(inp)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng(inp)
    torch.cuda.synchronize()
    out["graph_ms"] = timed(g.replay)
    eager_codes = eng.dense_codes(eng(inp)).clone()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    out["graph_replay_equals_eager"] = bool(torch.equal(eng.dense_codes(), eager_codes))
    out["graph_int8_top_s"] = 2 * macs / (out["graph_ms"] * 1e-3) / 1e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
