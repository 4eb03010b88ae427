"""TEST INFRASTRUCTURE: plain-Python statement of the pillar voxelizer contract (small clouds only).

The reference voxelizes with spconv's ``Point2VoxelCPU3d`` (``pre_processor/sp_voxel_preprocessor.py:54-85``), an
un-vendored third-party dependency whose version is not pinned: parity is *unpinned*; this is the contract the synthetic
inputs and the GPU voxelizer follow.  One pass over the points in index order, exactly as a first-come voxelizer works."""
import numpy as np


def voxelize(points, lidar_range, voxel_size, max_points=32, max_voxels=70000):
    lo = np.asarray(lidar_range[:3], np.float32)
    vs = np.asarray(voxel_size, np.float32)
    n = np.round((np.asarray(lidar_range[3:], np.float64) - np.asarray(lidar_range[:3], np.float64)) / np.asarray(voxel_size, np.float64)).astype(np.int64)
    slot_of = {}
    feats, coords, nums = [], [], []
    for p in np.asarray(points, np.float32):
        c = np.floor((p[:3] - lo) / vs)
        if (c < 0).any() or (c >= n).any():
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        v = slot_of.get(key)
        if v is None:
            if len(feats) >= max_voxels:
                slot_of[key] = -1               # voxel budget exhausted: later points of this cell are dropped too
                continue
            v = slot_of[key] = len(feats)
            feats.append(np.zeros((max_points, points.shape[1]), np.float32)); coords.append(key); nums.append(0)
        if v < 0:
            continue
        if nums[v] < max_points:
            feats[v][nums[v]] = p
            nums[v] += 1
    if not feats:
        return np.zeros((0, max_points, points.shape[1]), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32)
    return np.stack(feats), np.asarray(coords, np.int32), np.asarray(nums, np.int32)
