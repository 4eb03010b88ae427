"""Time the deployed Pyramid path at the V2X-Real grid (704 x 200 pillars -> 352 x 100 maps): whole frame as a HIP graph, then the agent
side (encode_features) and the ego side (decode_features) separately.  ``python tools/bench_pyramid.py [n_agents] [iters]``."""
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantv2x_amd import synth  # noqa: E402


def main():
    n_agents = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    shape = sys.argv[3] if len(sys.argv) > 3 else "v2xreal"
    import copy
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import train_utils
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax, wrap
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(32)
    hy = synth.make_pyramid_hypes(shape)
    model = train_utils.create_model(copy.deepcopy(hy)).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    t0 = time.time()
    calib = synth.scene_to_torch(synth.make_scene(shape, n_agents=1, seed=3, n_points=60000))
    st = export_ptq_state(calibrate_minmax(wrap(model), [calib]))          # the reference's min-max recipe: one observer pass, frozen
    print("calibration s", round(time.time() - t0, 1), flush=True)
    eng = deploy(state=st)
    sc = synth.make_scene(shape, n_agents=n_agents, seed=0)
    dd = synth.scene_to_torch(sc, "cuda")
    out = eng(dd)
    torch.cuda.synchronize()
    hw = eng.fh * eng.fw
    pw = dd["pairwise_t_matrix"].to(torch.float64).contiguous()

    def timeit(fn, k=iters):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    res = {"n_agents": n_agents, "pillars": int(dd["inputs_m1"]["voxel_features"].shape[0]), "map": [eng.fh, eng.fw]}
    replay = eng.capture(dd)
    res["frame_graph_ms"] = timeit(replay)
    res["frame_eager_ms"] = timeit(lambda: eng(dd))
    codes = eng.encode_features(dd["inputs_m1"], n_agents).clone()
    res["encode_features_ms"] = timeit(lambda: eng.encode_features(dd["inputs_m1"], n_agents))
    res["decode_features_ms"] = timeit(lambda: eng.decode_features(codes, hw, n_agents * hw, [n_agents], pw))
    res["agent_backbone_ms"] = timeit(lambda: eng.agent_backbone(n_agents))
    res["codebook_encode_ms"] = timeit(lambda: eng.encode_codes(n_agents))
    # throughput: batches of `frames` scenes (n_agents each) per call, one HIP graph per batch
    for frames in (2, 4):
        scenes = [synth.make_scene(shape, n_agents=n_agents, seed=3 + f) for f in range(frames)]
        parts = []
        for f, s in enumerate(scenes):
            part = {k: v.copy() for k, v in s["inputs_m1"].items()}
            part["voxel_coords"][:, 0] += f * n_agents
            parts.append(part)
        batch = {"inputs_m1": {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).cuda() for k in parts[0]},
                 "agent_modality_list": ["m1"] * (n_agents * frames), "record_len": torch.full((frames,), n_agents, dtype=torch.int64),
                 "pairwise_t_matrix": torch.from_numpy(np.concatenate([s["pairwise_t_matrix"] for s in scenes])).cuda()}
        rep = eng.capture(batch)
        ms = timeit(rep, max(10, iters // frames))
        res[f"batch{frames}_ms_per_frame"] = ms / frames
        res[f"batch{frames}_frames_per_s"] = 1000.0 * frames / ms
    print(json.dumps(res))


if __name__ == "__main__":
    main()
