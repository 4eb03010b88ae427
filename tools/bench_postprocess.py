"""Device time of qv2x_postprocess_f32 at V2X-Real head-map size (dev tool, runs on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import postprocess as P
from test_hip_postprocess import _params, _planted_scene
from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor

lidar = [-140.8, -40.0, -3.0, 140.8, 40.0, 1.0]
anchors = P.generate_anchor_box(lidar, 704, 200, 0.4, 0.4)
for n_obj in (20, 150, 600):
    cls, reg, dirp = _planted_scene(np.random.default_rng(5), 100, 352, n_obj)
    pp = build_postprocessor(_params(lidar, 704, 200), train=False)
    out = {"ego": {"cls_preds": torch.from_numpy(cls).cuda(), "reg_preds": torch.from_numpy(reg).cuda(), "dir_preds": torch.from_numpy(dirp).cuda()}}
    data = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": torch.from_numpy(anchors)}}
    for _ in range(3): b, s = pp.post_process(data, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): b, s = pp.post_process(data, out)
    e1.record(); torch.cuda.synchronize()
    print(f"{n_obj} planted objects: {len(s)} boxes out, {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per frame (incl. the count read-back)")
