"""qv2x_postprocess_f32 (through the mirrored VoxelPostprocessor class) vs oracle/postprocess.py.  Needs an MI355X.

Tolerance: fp32 expf / sinf / cosf of the device library vs numpy differ by an ulp or two -> corners and scores within
1e-5; the SET of boxes (threshold, filters, NMS decisions, range mask) and their order must be identical."""
import os

import numpy as np
import pytest
import torch

from oracle import postprocess as P

pytestmark = pytest.mark.gpu
TOL = dict(rtol=0, atol=1e-5)


def _params(lidar, grid_w, grid_h, thr=0.2):
    return {"core_method": "VoxelPostprocessor", "gt_range": list(lidar), "order": "hwl", "max_num": 100, "nms_thresh": 0.15,
            "anchor_args": {"cav_lidar_range": list(lidar), "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90], "feature_stride": 2, "num": 2,
                            "vw": 0.4, "vh": 0.4, "vd": 4.0, "W": grid_w, "H": grid_h, "D": 1},
            "target_args": {"score_threshold": thr}, "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]}}


def _run_gpu(params, cls, reg, dirp, anchors, t, **kw):
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    pp = build_postprocessor(params, train=False)
    dev = "cuda"
    out = {"ego": {"cls_preds": torch.from_numpy(cls).to(dev), "reg_preds": torch.from_numpy(reg).to(dev)}}
    if dirp is not None:
        out["ego"]["dir_preds"] = torch.from_numpy(dirp).to(dev)
    data = {"ego": {"transformation_matrix": torch.from_numpy(t), "anchor_box": torch.from_numpy(anchors)}}
    boxes, scores = pp.post_process(data, out, **kw)
    if boxes is None:
        return None, None
    return boxes.cpu().numpy(), scores.cpu().numpy()


@pytest.mark.parametrize("tag", ["ident", "moved"])
def test_golden_inputs_match_oracle(tag):
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess.npz")) as z:
        g = {k: z[k] for k in z.files}
    t = np.eye(4, dtype=np.float32) if tag == "ident" else g["T"]
    want_b, want_s = P.post_process(g["cls"], g["reg"], g["dir"], g["anchors"], t, g["lidar_range"])
    got_b, got_s = _run_gpu(_params(g["lidar_range"], 64, 32), g["cls"], g["reg"], g["dir"], g["anchors"], t)
    assert got_b.shape == want_b.shape and 0 < len(want_s) < len(g[tag + "_scores"])      # the NMS removed something
    np.testing.assert_allclose(got_s, want_s, **TOL)
    np.testing.assert_allclose(got_b, want_b, **TOL)


def _planted_scene(rng, h, w, n_obj):
    """background logits far below the threshold + clusters of overlapping detections around planted objects"""
    cls = rng.normal(-6.0, 0.8, size=(1, 2, h, w)).astype(np.float32)
    reg = rng.normal(0.0, 0.05, size=(1, 14, h, w)).astype(np.float32)
    dirp = rng.normal(0.0, 1.0, size=(1, 4, h, w)).astype(np.float32)
    for _ in range(n_obj):
        y, x, a = int(rng.integers(2, h - 2)), int(rng.integers(2, w - 2)), int(rng.integers(0, 2))
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                cls[0, a, y + dy, x + dx] = rng.normal(1.5, 1.0)
                reg[0, a * 7 + 0, y + dy, x + dx] = -dx * 0.8 / 4.2 + rng.normal(0, 0.02)     # regress back towards the centre cell
                reg[0, a * 7 + 1, y + dy, x + dx] = -dy * 0.8 / 4.2 + rng.normal(0, 0.02)
    return cls, reg, dirp


def test_v2xreal_size_planted_objects():
    """100 x 352 head maps (V2X-Real feature map), 70 400 anchors, 150 planted objects of nine detections each."""
    rng = np.random.default_rng(5)
    lidar = [-140.8, -40.0, -3.0, 140.8, 40.0, 1.0]
    h, w = 100, 352
    anchors = P.generate_anchor_box(lidar, 704, 200, 0.4, 0.4)
    cls, reg, dirp = _planted_scene(rng, h, w, 150)
    t = np.eye(4, dtype=np.float32)
    want_b, want_s = P.post_process(cls, reg, dirp, anchors, t, lidar)
    got_b, got_s = _run_gpu(_params(lidar, 704, 200), cls, reg, dirp, anchors, t)
    assert 100 <= len(want_s) <= 160
    assert got_b.shape == want_b.shape
    np.testing.assert_allclose(got_s, want_s, **TOL)
    np.testing.assert_allclose(got_b, want_b, **TOL)


def test_edge_cases():
    rng = np.random.default_rng(9)
    lidar = [-12.8, -6.4, -3.0, 12.8, 6.4, 1.0]
    anchors = P.generate_anchor_box(lidar, 64, 32, 0.4, 0.4)
    h, w = 16, 32
    t = np.eye(4, dtype=np.float32)
    # nothing above the threshold -> (None, None), as the reference
    cls = np.full((1, 2, h, w), -9.0, np.float32)
    reg = np.zeros((1, 14, h, w), np.float32)
    assert _run_gpu(_params(lidar, 64, 32), cls, reg, None, anchors, t) == (None, None)
    # no direction head; every anchor passes; the top-k cap (reference: 1000) applies before the NMS
    cls = rng.normal(2.0, 1.0, size=(1, 2, h, w)).astype(np.float32)
    reg = rng.normal(0.0, 0.05, size=(1, 14, h, w)).astype(np.float32)
    want_b, want_s = P.post_process(cls, reg, None, anchors, t, lidar)
    got_b, got_s = _run_gpu(_params(lidar, 64, 32), cls, reg, None, anchors, t)
    assert got_b.shape == want_b.shape
    np.testing.assert_allclose(got_s, want_s, **TOL)
    np.testing.assert_allclose(got_b, want_b, **TOL)
    # equal scores: the stable order (h, w, anchor) decides, on both sides
    cls[:] = 1.25
    want_b, want_s = P.post_process(cls, reg, None, anchors, t, lidar)
    got_b, got_s = _run_gpu(_params(lidar, 64, 32), cls, reg, None, anchors, t)
    assert got_b.shape == want_b.shape
    np.testing.assert_allclose(got_b, want_b, **TOL)


def test_v2xreal_size_thousands_of_equal_scores():
    """more equal top scores than the top-k selection's LDS sort holds (postprocess.hip: SEL_CAP = 2048): the exact slow path -- the stable
    (h, w, anchor) order decides which 1000 of them enter the NMS, as on the oracle's side"""
    rng = np.random.default_rng(11)
    lidar = [-140.8, -40.0, -3.0, 140.8, 40.0, 1.0]
    h, w = 100, 352
    anchors = P.generate_anchor_box(lidar, 704, 200, 0.4, 0.4)
    t = np.eye(4, dtype=np.float32)
    reg = rng.normal(0.0, 0.05, size=(1, 14, h, w)).astype(np.float32)
    cls = rng.normal(-1.0, 0.1, size=(1, 2, h, w)).astype(np.float32)              # above the threshold of 0.2, below the ties
    ties = rng.choice(h * w, size=3000, replace=False)
    cls.reshape(2, h * w)[0, ties] = 2.5                                            # 3000 equal scores at scattered cells: 1000 of them make the cut
    want_b, want_s = P.post_process(cls, reg, None, anchors, t, lidar)
    got_b, got_s = _run_gpu(_params(lidar, 704, 200), cls, reg, None, anchors, t)
    assert len(want_s) > 100 and got_b.shape == want_b.shape
    np.testing.assert_allclose(got_s, want_s, **TOL)
    np.testing.assert_allclose(got_b, want_b, **TOL)
    # a quantized head: eight different logits over all 70 400 anchors -- thousands of ties at every level, the cut inside one of them
    levels = np.array([-0.5, 0.0, 0.4, 0.9, 1.3, 1.8, 2.2, 3.0], np.float32)
    cls = levels[rng.integers(0, 8, size=(1, 2, h, w))]
    far = dict(rtol=0, atol=2e-5)                                      # (boxes out to |x| = 140 m: an fp32 ulp there is 1.5e-5)
    want_b, want_s = P.post_process(cls, reg, None, anchors, t, lidar)
    got_b, got_s = _run_gpu(_params(lidar, 704, 200), cls, reg, None, anchors, t)
    assert len(want_s) > 100 and got_b.shape == want_b.shape
    np.testing.assert_allclose(got_s, want_s, **TOL)
    np.testing.assert_allclose(got_b, want_b, **far)
    # every anchor the same score: the first 1000 slots enter
    cls[:] = 1.0
    want_b, want_s = P.post_process(cls, reg, None, anchors, t, lidar)
    got_b, got_s = _run_gpu(_params(lidar, 704, 200), cls, reg, None, anchors, t)
    assert got_b.shape == want_b.shape
    np.testing.assert_allclose(got_b, want_b, **far)


def test_errors():
    import ctypes as C
    from quantv2x_amd import lib as L
    lib = L.load()
    d = L.PostprocessDesc()
    d.h, d.w, d.anchors_per_cell, d.max_boxes, d.score_threshold = 16, 32, 2, 5000, 0.2
    d.num_classes, d.max_extent, d.z_min, d.z_max = 1, 6.0, -3.0, 1.0
    assert lib.qv2x_postprocess_workspace_bytes(C.byref(d)) == -1 and b"max_boxes" in lib.qv2x_last_error()
    d.max_boxes = 1000
    need = lib.qv2x_postprocess_workspace_bytes(C.byref(d))
    assert need > 0
    x = torch.zeros(1024, dtype=torch.float32, device="cuda")
    rc = lib.qv2x_postprocess_f32(C.byref(d), L.ptr(x), L.ptr(x), None, L.ptr(x), L.ptr(x), 16, L.ptr(x), L.ptr(x), None, L.ptr(x), None)
    assert rc != 0 and b"workspace" in lib.qv2x_last_error()
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    with pytest.raises(NotImplementedError):
        build_postprocessor({"core_method": "BevPostprocessor"}, train=False)
    pp = build_postprocessor(_params([-12.8, -6.4, -3.0, 12.8, 6.4, 1.0], 64, 32), train=False)
    with pytest.raises(RuntimeError):                                   # CPU tensors: no fallback
        pp.post_process({"ego": {}}, {"ego": {"cls_preds": torch.zeros(1, 2, 16, 32), "reg_preds": torch.zeros(1, 14, 16, 32)}})


# ---- multi-class (VoxelPostprocessor3Heads) ------------------------------------------------------------------------------
def _run_gpu_mc(params, cls, reg, all_anchors, t):
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    pp = build_postprocessor(params, train=False)
    out = {"ego": {"cls_preds": torch.from_numpy(cls).cuda(), "reg_preds": torch.from_numpy(reg).cuda()}}
    data = {"ego": {"transformation_matrix": torch.from_numpy(t), "all_anchors": torch.from_numpy(all_anchors), "num_anchors_per_location": [2, 2, 2]}}
    boxes, sl = pp.post_process(data, out)
    return (None, None) if boxes is None else (boxes.cpu().numpy(), sl.cpu().numpy())


def _oracle_mc(cls, reg, all_anchors, t, gt_range):
    from test_postprocess_oracle import interleave
    return P.post_process(cls, reg, None, interleave(all_anchors), t, gt_range, num_classes=3, max_extent=100.0, z_lim=(-100.0, 100.0),
                          range_xy_only=True, return_labels=True)


@pytest.mark.parametrize("tag", ["ident", "moved"])
def test_mc_golden_inputs_match_oracle(tag):
    from test_postprocess_oracle import mc_params
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess_mc.npz")) as z:
        g = {k: z[k] for k in z.files}
    t = np.eye(4, dtype=np.float32) if tag == "ident" else g["T"]
    wb, ws, wl = _oracle_mc(g["cls"], g["reg"], g["all_anchors"], t, g["gt_range"])
    gb, gsl = _run_gpu_mc(mc_params(g["lidar_range"], 64, 32), g["cls"], g["reg"], g["all_anchors"], t)
    assert gb.shape == wb.shape and 0 < len(ws) < len(g[tag + "_boxes"])
    np.testing.assert_array_equal(gsl[:, 1].astype(np.int64), wl)
    np.testing.assert_allclose(gsl[:, 0], ws, **TOL)
    np.testing.assert_allclose(gb, wb, **TOL)


def test_mc_late_fusion_two_cavs_match_oracle():
    """late fusion on the GPU (qv2x_postprocess_late_f32 behind the plugin class): two CAVs with their own head maps and matrices, NMS
    over the union -- identical box set and order as the oracle"""
    from test_postprocess_oracle import interleave, mc_params
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess_mc.npz")) as z:
        g = {k: z[k] for k in z.files}
    anchors = interleave(g["all_anchors"])
    cavs = [(g["cls"], g["reg"], None, anchors, np.eye(4, dtype=np.float32)), (g["late_cls2"], g["late_reg2"], None, anchors, g["T"])]
    wb, ws, wl = P.post_process_late(cavs, g["gt_range"], num_classes=3, max_extent=100.0, z_lim=(-100.0, 100.0), range_xy_only=True, return_labels=True)
    pp = build_postprocessor(mc_params(g["lidar_range"], 64, 32), train=False)
    aa = torch.from_numpy(g["all_anchors"])
    data = {"ego": {"transformation_matrix": torch.eye(4), "all_anchors": aa, "num_anchors_per_location": [2, 2, 2]},
            "cav1": {"transformation_matrix": torch.from_numpy(g["T"]), "all_anchors": aa, "num_anchors_per_location": [2, 2, 2]}}
    out = {"ego": {"cls_preds": torch.from_numpy(g["cls"]).cuda(), "reg_preds": torch.from_numpy(g["reg"]).cuda()},
           "cav1": {"cls_preds": torch.from_numpy(g["late_cls2"]).cuda(), "reg_preds": torch.from_numpy(g["late_reg2"]).cuda()}}
    boxes, sl = pp.post_process(data, out)
    gb, gsl = boxes.cpu().numpy(), sl.cpu().numpy()
    assert gb.shape == wb.shape and len(wb) > 0
    np.testing.assert_array_equal(gsl[:, 1].astype(np.int64), wl)
    np.testing.assert_allclose(gsl[:, 0], ws, **TOL)
    np.testing.assert_allclose(gb, wb, **TOL)
    single, _ = pp.post_process({"ego": data["ego"]}, {"ego": out["ego"]})          # one CAV through the same entry still works
    assert 0 < len(single) < len(gb) + 1


def test_mc_deployed_model_heads_to_boxes():
    """The multi-class deployed model's own head maps (tiny shape, 3 agents) through the GPU post-processor == the oracle
    post-processor on the same maps: the frame ends in boxes without leaving the GPU except for the box count."""
    from _common import calibrated_plugin, scene_np
    from test_postprocess_oracle import MC_CFGS, mc_params
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    eng = deploy(state=export_ptq_state(calibrated_plugin()))
    out = eng(synth.scene_to_torch(scene_np(3), "cuda"))
    lidar, vox = synth.SHAPES["tiny"][0], synth.SHAPES["tiny"][1]
    gw, gh, _ = synth.grid_size(lidar, vox)
    aa, _ = P.generate_anchor_boxes_3heads(lidar, gw, gh, MC_CFGS)
    all_anchors = np.array(aa)
    cls, reg = out["cls_preds"].cpu().numpy(), out["reg_preds"].cpu().numpy()
    assert cls.shape[1] == 18 and reg.shape[1] == 42
    t = np.eye(4, dtype=np.float32)
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    pp = build_postprocessor(mc_params(lidar, gw, gh), train=False)
    data = {"ego": {"transformation_matrix": torch.from_numpy(t), "all_anchors": torch.from_numpy(all_anchors), "num_anchors_per_location": [2, 2, 2]}}
    boxes, sl = pp.post_process(data, {"ego": out})
    wb, ws, wl = _oracle_mc(cls, reg, all_anchors, t, pp.gt_range)
    if boxes is None:
        assert len(ws) == 0
        return
    assert boxes.shape[0] == len(ws)
    np.testing.assert_allclose(sl[:, 0].cpu().numpy(), ws, **TOL)
    np.testing.assert_allclose(boxes.cpu().numpy(), wb, **TOL)
