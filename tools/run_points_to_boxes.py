"""Whole chain on the GPU for one V2X-Real frame: LiDAR sweep -> pillars (qv2x_voxelize_f32) -> deployed W8A8 model ->
boxes (qv2x_postprocess_f32).  Dev tool, runs on the GPU box: prints wall-clock per frame of each part (eager launches,
two host read-backs per frame: the pillar count and the box count)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
from quantv2x_amd.voxelizer import GpuVoxelizer
from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
from test_postprocess_oracle import MC_CFGS, mc_params

shape = "v2xreal"
lidar, vox, max_vox, _ = synth.SHAPES[shape]
eng = deploy(path=os.path.join(ROOT, "tools", "cache", "v2xreal_state.npz"))
vz = GpuVoxelizer(lidar, vox, 32, max_vox)
gw, gh, _ = synth.grid_size(lidar, vox)
pp = build_postprocessor(mc_params(lidar, gw, gh), train=False)
all_anchors, per_loc = pp.generate_anchor_box()
all_anchors = torch.from_numpy(np.array(all_anchors))
pts = torch.from_numpy(synth.make_points(lidar, 60000, 3000)).cuda()
pairwise = torch.eye(4, dtype=torch.float64).reshape(1, 1, 1, 4, 4).repeat(1, 5, 5, 1, 1).cuda()
cav = {"transformation_matrix": torch.eye(4), "all_anchors": all_anchors, "num_anchors_per_location": per_loc}


def frame():
    t0 = time.perf_counter()
    inputs = vz([pts])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = eng({"inputs_m1": inputs, "agent_modality_list": ["m1"], "record_len": torch.tensor([1]), "pairwise_t_matrix": pairwise})
    torch.cuda.synchronize(); t2 = time.perf_counter()
    boxes, score_labels = pp.post_process({"ego": cav}, {"ego": out})
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2), inputs["voxel_features"].shape[0], 0 if boxes is None else boxes.shape[0]


for _ in range(5): frame()
acc = np.zeros(3); n = 30
for _ in range(n):
    t, m, k = frame(); acc += t
acc *= 1e3 / n
print(f"{m} pillars, {k} boxes (random-weight model: the boxes are noise, the work is real)")
print(f"voxelize {acc[0]:.3f} ms | model (eager launches) {acc[1]:.3f} ms | post-process {acc[2]:.3f} ms | total {acc.sum():.3f} ms per frame")
