"""Dev tool (GPU box): the post-process (f2) and the voxelizer (f1) of ONE V2X-Real frame, 50 eager repetitions each -- for rocprofv3 --kernel-trace --stats.
    python tools/bench_post.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from quantv2x_amd import lib as _L
TAG = sys.argv[1] if len(sys.argv) > 1 else ""
if TAG:
    _L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{TAG}.so")
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
from quantv2x_amd.plugin.data_utils.post_processor.voxel_postprocessor import gpu_post_process
from quantv2x_amd.voxelizer import GpuVoxelizer
device = torch.device("cuda", 0)
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
lidar, vox, max_vox, _ = synth.SHAPES[bench.SHAPE]
gw, gh, _ = synth.grid_size(lidar, vox)
vz = GpuVoxelizer(lidar, vox, 32, max_vox)
pp = build_postprocessor(synth.mc_postprocess_params(lidar, gw, gh), train=False)
all_anchors, _ = pp.generate_anchor_box()
a = torch.as_tensor(np.array(all_anchors)).to(torch.float32).permute(1, 2, 0, 3, 4).contiguous()
anchors_dev, per_cell = a.reshape(-1, 7).to(device), int(a.shape[2] * a.shape[3])
sweeps = [torch.from_numpy(synth.make_points(lidar, bench.N_POINTS, 3000)).to(device)]
pairwise = torch.eye(4, dtype=torch.float64).reshape(1, 1, 1, 4, 4).repeat(1, 5, 5, 1, 1).to(device)
inp = vz.fixed(sweeps, 40960)
o = eng({"inputs_m1": inp, "agent_modality_list": ["m1"], "record_len": torch.ones(1, dtype=torch.int64), "pairwise_t_matrix": pairwise})
torch.cuda.synchronize()


def post():
    return gpu_post_process(pp, o["cls_preds"][:1], o["reg_preds"][:1], None, anchors_dev, torch.eye(4), anchors_per_cell=per_cell,
                            num_classes=int(o["cls_preds"].shape[1] // per_cell), num_bins=0, dir_offset=0.0, rng=pp.gt_range, range_xy_only=True,
                            max_extent=100.0, z_lim=(-100.0, 100.0), max_boxes=1000, sync=False)
for _ in range(50):
    r = post()
torch.cuda.synchronize()
print("boxes", int(r[3].item()))
for _ in range(50):
    vz.fixed(sweeps, 40960)
torch.cuda.synchronize()
print(f"post-process in a graph: {bench.event_time_us(bench._graph_of(post), 20):.1f} us")
# the same maps with the class logits snapped to eight levels (a coarsely quantized head: thousands of equal scores at the top-k cut)
lv = torch.tensor([-0.5, 0.0, 0.4, 0.9, 1.3, 1.8, 2.2, 3.0], device=device)
o["cls_preds"] = lv[torch.bucketize(o["cls_preds"].contiguous(), (lv[1:] + lv[:-1]) / 2)]
print(f"post-process with eight-level class logits (the exact tie path of the selection): {bench.event_time_us(bench._graph_of(post), 20):.1f} us, boxes {int(post()[3].item())}")
print(f"voxelizer in a graph: {bench.event_time_us(bench._graph_of(lambda: vz.fixed(sweeps, 40960)), 20):.1f} us")
if TAG:
    import ctypes as C
    buf = np.zeros(8, np.int64)
    if C.CDLL(_L.LIB_PATH).qv2x_debug_pp_fine(buf.ctypes.data_as(C.c_void_p)) == 0:
        print("sweep kernel phases (ticks): copy to LDS", buf[1] - buf[0], "| wave loop", buf[2] - buf[1], "| barrier", buf[3] - buf[2], "| range mask + prefix", buf[4] - buf[3], "| output", buf[5] - buf[4])
