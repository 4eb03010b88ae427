"""f2 post-processing: the CPU oracle against vectors captured from the reference's own
``VoxelPostprocessor.post_process`` (tests/golden/make_golden.py::gen_postprocess; its shapely NMS replaced by
"keep all, score order" there), plus the parts the reference cannot pin here (polygon IoU, greedy NMS)."""
import os

import numpy as np
import pytest

from oracle import postprocess as P

GOLD = os.path.join(os.path.dirname(__file__), "golden", "postprocess.npz")
TOL = dict(rtol=0, atol=2e-6)          # fp32 exp / sin / cos of numpy vs torch differ by an ulp


@pytest.fixture(scope="module")
def gold():
    with np.load(GOLD) as z:
        return {k: z[k] for k in z.files}


def test_anchor_boxes_equal_reference(gold):
    a = P.generate_anchor_box(gold["lidar_range"], 64, 32, 0.4, 0.4)
    np.testing.assert_array_equal(a, gold["anchors"])
    from quantv2x_amd.plugin.data_utils.post_processor import VoxelPostprocessor
    lr = [float(v) for v in gold["lidar_range"]]
    pp = VoxelPostprocessor({"order": "hwl", "anchor_args": {"cav_lidar_range": lr, "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90],
                                                             "feature_stride": 2, "num": 2, "vw": 0.4, "vh": 0.4, "W": 64, "H": 32}}, train=False)
    np.testing.assert_array_equal(pp.generate_anchor_box(), gold["anchors"])


def test_delta_to_boxes3d(gold):
    np.testing.assert_allclose(P.delta_to_boxes3d(gold["reg"], gold["anchors"]), gold["delta_to_boxes3d"][0], **TOL)


@pytest.mark.parametrize("tag", ["ident", "moved"])
def test_flow_without_nms_matches_reference(gold, tag):
    t = np.eye(4, dtype=np.float32) if tag == "ident" else gold["T"]
    boxes, scores = P.post_process(gold["cls"], gold["reg"], gold["dir"], gold["anchors"], t, gold["lidar_range"], nms=False)
    assert boxes.shape == gold[tag + "_boxes"].shape            # same candidates survive threshold, filters and range mask
    np.testing.assert_allclose(scores, gold[tag + "_scores"], **TOL)
    np.testing.assert_allclose(boxes, gold[tag + "_boxes"], **TOL)


def test_polygon_intersection():
    sq = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], float)
    assert P.quad_intersection_area(sq, sq + 1) == pytest.approx(1.0)
    assert P.quad_intersection_area(sq, sq[::-1]) == pytest.approx(4.0)             # orientation does not matter
    assert P.quad_intersection_area(sq, sq + 5) == 0.0
    rot = np.array([[1, -0.41421356], [2.41421356, 1], [1, 2.41421356], [-0.41421356, 1]])    # same square turned by 45 deg
    assert P.quad_intersection_area(sq, rot) == pytest.approx(8 * (np.sqrt(2) - 1), rel=1e-6)  # regular octagon


def test_nms_is_greedy_in_score_order(gold):
    boxes, scores = P.post_process(gold["cls"], gold["reg"], gold["dir"], gold["anchors"], np.eye(4, dtype=np.float32),
                                   gold["lidar_range"], nms=False)
    keep = P.nms_rotated(boxes, scores, 0.15)
    assert 0 < len(keep) < len(scores) and keep[0] == 0
    quads = boxes[:, :4, :2].astype(np.float64)
    for a_i, i in enumerate(keep):                                   # kept boxes do not overlap each other beyond the threshold
        for j in keep[a_i + 1:]:
            inter = P.quad_intersection_area(quads[i], quads[j])
            assert inter / (P.quad_area(quads[i]) + P.quad_area(quads[j]) - inter) <= 0.15
    dropped = sorted(set(range(len(scores))) - set(keep.tolist()))
    for j in dropped:                                                # every dropped box overlaps a kept, better one
        assert any(i < j and P.quad_intersection_area(quads[i], quads[j]) /
                   (P.quad_area(quads[i]) + P.quad_area(quads[j]) - P.quad_intersection_area(quads[i], quads[j])) > 0.15 for i in keep)


# ---- multi-class (VoxelPostprocessor3Heads, the V2X-Real yaml) ---------------------------------------------------------
from quantv2x_amd.synth import MC_ANCHOR_CFGS as MC_CFGS, mc_postprocess_params as mc_params  # noqa: E402  (shared with bench.py)


@pytest.fixture(scope="module")
def gold_mc():
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess_mc.npz")) as z:
        return {k: z[k] for k in z.files}


def interleave(all_anchors):
    """(num_class, H, W, A, 7) -> (H, W, num_class * A, 7), voxel_postprocessor_3heads.py:354-358"""
    a = np.transpose(all_anchors, (1, 2, 0, 3, 4))
    return a.reshape(a.shape[0], a.shape[1], -1, 7)


def test_mc_anchor_boxes_equal_reference(gold_mc):
    aa, per_loc = P.generate_anchor_boxes_3heads(gold_mc["lidar_range"], 64, 32, MC_CFGS)
    np.testing.assert_array_equal(np.array(aa), gold_mc["all_anchors"])
    assert per_loc == list(gold_mc["num_anchors_per_location"])
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    pp = build_postprocessor(mc_params([float(v) for v in gold_mc["lidar_range"]], 64, 32), train=False)
    got, per_loc2 = pp.generate_anchor_box()
    np.testing.assert_array_equal(np.array(got), gold_mc["all_anchors"])
    assert per_loc2 == per_loc and pp.gt_range == [float(v) for v in gold_mc["gt_range"]]


@pytest.mark.parametrize("tag", ["ident", "moved"])
def test_mc_flow_without_nms_matches_reference(gold_mc, tag):
    t = np.eye(4, dtype=np.float32) if tag == "ident" else gold_mc["T"]
    boxes, scores, labels = P.post_process(gold_mc["cls"], gold_mc["reg"], None, interleave(gold_mc["all_anchors"]), t, gold_mc["gt_range"],
                                           nms=False, num_classes=3, max_extent=100.0, z_lim=(-100.0, 100.0), range_xy_only=True,
                                           return_labels=True)
    want = gold_mc[tag + "_score_labels"]
    assert boxes.shape == gold_mc[tag + "_boxes"].shape
    np.testing.assert_array_equal(labels, want[:, 1].astype(np.int64))
    np.testing.assert_allclose(scores, want[:, 0], **TOL)
    np.testing.assert_allclose(boxes, gold_mc[tag + "_boxes"], **TOL)


def test_mc_late_fusion_without_nms_matches_reference(gold_mc):
    """two CAVs in one ``post_process`` call (late fusion, voxel_postprocessor_3heads.py:345-420): the union in the reference's order"""
    anchors = interleave(gold_mc["all_anchors"])
    cavs = [(gold_mc["cls"], gold_mc["reg"], None, anchors, np.eye(4, dtype=np.float32)),
            (gold_mc["late_cls2"], gold_mc["late_reg2"], None, anchors, gold_mc["T"])]
    boxes, scores, labels = P.post_process_late(cavs, gold_mc["gt_range"], nms=False, num_classes=3, max_extent=100.0, z_lim=(-100.0, 100.0),
                                                range_xy_only=True, return_labels=True)
    want = gold_mc["late_score_labels"]
    assert boxes.shape == gold_mc["late_boxes"].shape and len(boxes) > len(gold_mc["ident_boxes"])
    np.testing.assert_array_equal(labels, want[:, 1].astype(np.int64))
    np.testing.assert_allclose(scores, want[:, 0], **TOL)
    np.testing.assert_allclose(boxes, gold_mc["late_boxes"], **TOL)
