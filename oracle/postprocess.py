"""CPU restatement of the anchor-head post-processing -- TEST INFRASTRUCTURE ONLY.

Follows ``VoxelPostprocessor.post_process`` (opencood/data_utils/post_processor/voxel_postprocessor.py:245-405) for
one CAV (intermediate fusion: ``data_dict`` holds the ego only):

  anchors           generate_anchor_box                              voxel_postprocessor.py:30-83
  scores            sigmoid(cls.permute(0, 2, 3, 1)).reshape(-1)     :289-291, threshold :302-304
  boxes             delta_to_boxes3d                                 :408-453
  direction fix     limit_period(yaw - off, 0, pi) + off + pi*label  :316-331, common_utils.py:104-113
  corners           boxes_to_corners_3d (order 'hwl')                box_utils.py:152-204, common_utils.py:139-161
  projection        project_box3d                                    box_utils.py:278-316
  size / z filters  remove_large_pred_bbx (its z extent is computed from y, and only tested for != 0: kept),
                    remove_bbx_abnormal_z                            box_utils.py:916-966
  NMS               nms_rotated: top 1000 by score, greedy, IoU of the BOTTOM-face quadrilaterals > thresh removes
                                                                     box_utils.py:769-814
  range mask        mask_boxes_outside_range_numpy (all 8 corners)   box_utils.py:384-421

Pinned against the reference by ``tests/golden/postprocess.npz`` for everything except the polygon IoU itself: the
reference computes it with shapely (absent here, un-vendored: parity unpinned for that function); this file uses
Sutherland-Hodgman clipping of the two convex quadrilaterals in float64.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def generate_anchor_box(lidar_range, grid_w, grid_h, vw, vh, l=3.9, w=1.6, h=1.56, yaws_deg=(0, 90), feature_stride=2):
    """[H/stride, W/stride, A, 7] float64, order 'hwl': (x, y, z, h, w, l, yaw)."""
    r = [math.radians(e) for e in yaws_deg]
    x = np.linspace(lidar_range[0] + vw, lidar_range[3] - vw, grid_w // feature_stride)
    y = np.linspace(lidar_range[1] + vh, lidar_range[4] - vh, grid_h // feature_stride)
    cx, cy = np.meshgrid(x, y)
    a = len(r)
    cx = np.tile(cx[..., None], a)
    cy = np.tile(cy[..., None], a)
    cz = np.ones_like(cx) * -1.0
    r_ = np.ones_like(cx)
    for i in range(a):
        r_[..., i] = r[i]
    return np.stack([cx, cy, cz, np.ones_like(cx) * h, np.ones_like(cx) * w, np.ones_like(cx) * l, r_], axis=-1)


def generate_anchor_boxes_3heads(lidar_range, grid_w, grid_h, configs, order="hwl"):
    """VoxelPostprocessor3Heads.generate_anchor_box (voxel_postprocessor_3heads.py:63-132): one [H', W', R, 7] array per
    anchor set (class), and the anchors per location of each set."""
    out, per_loc = [], []
    for cfg in configs:
        gs = np.array([grid_w, grid_h]) // cfg["feature_map_stride"]
        size, rot, height = cfg["anchor_sizes"], cfg["anchor_rotations"], cfg["anchor_bottom_heights"]
        per_loc.append(len(rot) * len(size) * len(height))
        if cfg.get("align_center", False):
            xs, ys = (lidar_range[3] - lidar_range[0]) / gs[0], (lidar_range[4] - lidar_range[1]) / gs[1]
            xo, yo = xs / 2, ys / 2
        else:
            xs, ys = (lidar_range[3] - lidar_range[0]) / (gs[0] - 1), (lidar_range[4] - lidar_range[1]) / (gs[1] - 1)
            xo, yo = 0, 0
        x = np.arange(lidar_range[0] + xo, lidar_range[3] + 1e-5, step=xs)
        y = np.arange(lidar_range[1] + yo, lidar_range[4] + 1e-5, step=ys)
        z = np.array(height)
        rot, size = np.array(rot), np.array(size)
        xg, yg, zg = np.meshgrid(x, y, z)
        a = np.concatenate([xg, yg, zg], axis=-1)
        sz = np.tile(size.reshape(1, -1, 3), (*a.shape[0:2], 1))
        sz = sz[..., [2, 1, 0]] if order == "hwl" else sz[..., [0, 2, 1]]
        a = np.concatenate((a, sz), axis=-1)
        a = np.tile(a[:, :, None, :], (1, 1, len(rot), 1))
        r = np.tile(rot.reshape(1, 1, -1, 1), (*a.shape[0:2], len(size), 1))
        out.append(np.concatenate([a, r], axis=-1))
    return out, per_loc


def sigmoid(x):
    x = x.astype(F)
    return (F(1.0) / (F(1.0) + np.exp(-x, dtype=F))).astype(F)


def delta_to_boxes3d(reg, anchors):
    """reg f32 [1, 7A, H, W], anchors [H, W, A, 7] -> f32 [H*W*A, 7]."""
    d = np.transpose(reg, (0, 2, 3, 1)).reshape(-1, 7).astype(F)
    a = anchors.reshape(-1, 7).astype(F)
    diag = np.sqrt(a[:, 4] * a[:, 4] + a[:, 5] * a[:, 5], dtype=F)
    out = np.zeros_like(d)
    out[:, 0] = d[:, 0] * diag + a[:, 0]
    out[:, 1] = d[:, 1] * diag + a[:, 1]
    out[:, 2] = d[:, 2] * a[:, 3] + a[:, 2]
    out[:, 3:6] = np.exp(d[:, 3:6], dtype=F) * a[:, 3:6]
    out[:, 6] = d[:, 6] + a[:, 6]
    return out


def limit_period(val, offset, period):
    val = val.astype(F)
    return (val - np.floor(val / F(period) + F(offset)) * F(period)).astype(F)


def boxes_to_corners_3d(boxes):
    """'hwl' boxes [N, 7] -> corners f32 [N, 8, 3] (bottom face first, box_utils.py:152-204)."""
    b = boxes[:, [0, 1, 2, 5, 4, 3, 6]].astype(F)
    template = np.array([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                         [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]], dtype=F) / F(2)
    c = b[:, None, 3:6] * template[None]
    cosa, sina = np.cos(b[:, 6], dtype=F), np.sin(b[:, 6], dtype=F)
    x = c[:, :, 0] * cosa[:, None] + c[:, :, 1] * (-sina)[:, None]
    y = c[:, :, 0] * sina[:, None] + c[:, :, 1] * cosa[:, None]
    out = np.stack([x, y, c[:, :, 2]], axis=-1).astype(F)
    return out + b[:, None, 0:3]


def project_box3d(corners, t):
    t = t.astype(F)
    homo = np.concatenate([corners, np.ones(corners.shape[:2] + (1,), F)], axis=-1)      # [N, 8, 4]
    return np.einsum("ij,nkj->nki", t, homo).astype(F)[:, :, :3]


def quad_intersection_area(p, q):
    """Area of the intersection of two convex quadrilaterals (float64, any orientation): Sutherland-Hodgman."""
    def signed_area(poly):
        s = 0.0
        for i in range(len(poly)):
            x0, y0 = poly[i]
            x1, y1 = poly[(i + 1) % len(poly)]
            s += x0 * y1 - x1 * y0
        return 0.5 * s
    clip = [tuple(map(float, v)) for v in q]
    if signed_area(clip) < 0:
        clip = clip[::-1]
    out = [tuple(map(float, v)) for v in p]
    for i in range(4):
        ax, ay = clip[i]
        bx, by = clip[(i + 1) % 4]
        if not out:
            break
        inp, out = out, []
        for j in range(len(inp)):
            cx, cy = inp[j]
            dx, dy = inp[(j + 1) % len(inp)]
            sc = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
            sd = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax)
            if sc >= 0:
                out.append((cx, cy))
            if (sc >= 0) != (sd >= 0):
                t = sc / (sc - sd)
                out.append((cx + t * (dx - cx), cy + t * (dy - cy)))
    return abs(signed_area(out)) if len(out) >= 3 else 0.0


def quad_area(p):
    s = 0.0
    for i in range(4):
        s += float(p[i][0]) * float(p[(i + 1) % 4][1]) - float(p[(i + 1) % 4][0]) * float(p[i][1])
    return abs(0.5 * s)


def nms_rotated(corners, scores, thresh, top=1000):
    """Greedy rotated NMS on the bottom faces; returns indices in pick order.  Ties in score: lower index first."""
    order = np.argsort(-scores, kind="stable")[:top]
    quads = corners[:, :4, :2].astype(np.float64)
    areas = [quad_area(q) for q in quads]
    alive = list(order)
    pick = []
    while alive:
        i = alive.pop(0)
        pick.append(i)
        rest = []
        for j in alive:
            inter = quad_intersection_area(quads[i], quads[j])
            union = areas[i] + areas[j] - inter
            if not (union > 0 and inter / union > thresh):
                rest.append(j)
        alive = rest
    return np.array(pick, dtype=np.int64)


def _candidates(cls, reg, dirp, anchors, t, score_threshold, dir_offset, num_bins, num_classes):
    """one CAV: thresholded, decoded, direction-fixed boxes as corners projected by ``t`` (+ scores, labels), reference order"""
    if num_classes == 1:
        prob = sigmoid(np.transpose(cls, (0, 2, 3, 1))).reshape(-1)
        labels = np.ones(prob.shape, np.int64)
    else:
        pk = sigmoid(np.transpose(cls, (0, 2, 3, 1))).reshape(-1, num_classes)
        prob, labels = pk.max(axis=-1), pk.argmax(axis=-1) + 1
    boxes = delta_to_boxes3d(reg, anchors)
    mask = prob > F(score_threshold)
    boxes, scores, labels = boxes[mask], prob[mask], labels[mask]
    if boxes.shape[0] == 0:
        return np.zeros((0, 8, 3), F), np.zeros((0,), F), np.zeros((0,), np.int64)
    if dirp is not None:
        dm = np.transpose(dirp, (0, 2, 3, 1)).reshape(-1, num_bins)[mask]
        dl = np.argmax(dm, axis=-1).astype(F)
        period = 2 * np.pi / num_bins
        rot = limit_period(boxes[:, 6] - F(dir_offset), 0.0, period)
        boxes[:, 6] = rot + F(dir_offset) + F(period) * dl
        boxes[:, 6] = limit_period(boxes[:, 6], 0.5, 2 * np.pi)
    return project_box3d(boxes_to_corners_3d(boxes), t), scores, labels


def post_process_late(cavs, lidar_range, score_threshold=0.2, nms_thresh=0.15, dir_offset=0.7853, num_bins=2, nms=True, num_classes=1,
                      max_extent=6.0, z_lim=(-3.0, 1.0), range_xy_only=False, return_labels=False):
    """``cavs``: list of ``(cls, reg, dirp, anchors, t)``, one per CAV in the order of the reference's ``data_dict`` -- late fusion
    (voxel_postprocessor.py:272-345 / voxel_postprocessor_3heads.py:345-420): candidates of every CAV concatenated, then the single-CAV
    tail (size / z filters, NMS, range mask) over the union.  -> as ``post_process``."""
    parts = [_candidates(c, r, d, a, t, score_threshold, dir_offset, num_bins, num_classes) for (c, r, d, a, t) in cavs]
    corners = np.concatenate([p[0] for p in parts]); scores = np.concatenate([p[1] for p in parts]); labels = np.concatenate([p[2] for p in parts])
    if corners.shape[0] == 0:
        return (np.zeros((0, 8, 3), F), np.zeros((0,), F)) + ((np.zeros((0,), np.int64),) if return_labels else ())
    xl = corners[:, :, 0].max(1) - corners[:, :, 0].min(1)
    yl = corners[:, :, 1].max(1) - corners[:, :, 1].min(1)
    keep = (xl <= max_extent) & (yl <= max_extent) & (yl != 0)     # remove_large_pred_bbx, quirk kept
    keep &= (corners[:, :, 2].min(1) >= z_lim[0]) & (corners[:, :, 2].max(1) <= z_lim[1])
    corners, scores, labels = corners[keep], scores[keep], labels[keep]
    idx = nms_rotated(corners, scores, nms_thresh) if nms else np.argsort(-scores, kind="stable")
    corners, scores, labels = corners[idx], scores[idx], labels[idx]
    lo, hi = np.asarray(lidar_range[:3], F), np.asarray(lidar_range[3:], F)
    if range_xy_only:
        inside = ((corners[:, :, :2] >= lo[:2]) & (corners[:, :, :2] <= hi[:2])).all(axis=2).all(axis=1)
    else:
        inside = ((corners >= lo) & (corners <= hi)).all(axis=2).sum(axis=1) >= 8
    if return_labels:
        return corners[inside], scores[inside], labels[inside]
    return corners[inside], scores[inside]


def post_process(cls, reg, dirp, anchors, t, lidar_range, score_threshold=0.2, nms_thresh=0.15, dir_offset=0.7853,
                 num_bins=2, nms=True, num_classes=1, max_extent=6.0, z_lim=(-3.0, 1.0), range_xy_only=False, return_labels=False):
    """-> (corners f32 [K, 8, 3], scores f32 [K]) in descending score order; ``nms=False`` keeps every filtered box.

    ``num_classes > 1`` is VoxelPostprocessor3Heads.post_process (voxel_postprocessor_3heads.py:318-478): ``anchors`` is
    then [H, W, A_all, 7] with the anchor sets already interleaved per cell ((h, w, class set, rotation) order, :354-358),
    the score the largest class probability, no direction fix; box_utils_mc's limits are ``max_extent=100``,
    ``z_lim=(-100, 100)``, ``range_xy_only=True`` with ``lidar_range`` = GT_RANGE."""
    return post_process_late([(cls, reg, dirp, anchors, t)], lidar_range, score_threshold, nms_thresh, dir_offset, num_bins, nms, num_classes,
                             max_extent, z_lim, range_xy_only, return_labels)
