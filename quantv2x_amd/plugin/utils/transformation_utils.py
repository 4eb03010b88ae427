"""Pose helpers on the hot path; mirror of ``opencood/utils/transformation_utils.py``
(``normalize_pairwise_tfm :68-92``, ``get_pairwise_transformation :21-66``, ``x_to_world :264-310``)."""
import numpy as np


def normalize_pairwise_tfm(pairwise_t_matrix, H, W, discrete_ratio, downsample_rate=1):
    """[B, L, L, 4, 4] pairwise transforms -> [B, L, L, 2, 3] matrices for ``F.affine_grid``.

    Rows {0,1} x cols {0,1,3}; the off-diagonal terms are rescaled by the aspect ratio and the
    translation is expressed in half-extents (H, W in metres when discrete_ratio == 1)."""
    a = pairwise_t_matrix[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]]
    a[..., 0, 1] = a[..., 0, 1] * H / W
    a[..., 1, 0] = a[..., 1, 0] * W / H
    a[..., 0, 2] = a[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    a[..., 1, 2] = a[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return a


def x_to_world(pose):
    """[x, y, z, roll, yaw, pitch] (degrees) -> 4x4 T_world_x (CARLA convention)."""
    x, y, z, roll, yaw, pitch = pose[:]
    cy, sy = np.cos(np.radians(yaw)), np.sin(np.radians(yaw))
    cr, sr = np.cos(np.radians(roll)), np.sin(np.radians(roll))
    cp, sp = np.cos(np.radians(pitch)), np.sin(np.radians(pitch))
    return np.array([
        [cp * cy, cy * sp * sr - sy * cr, -cy * sp * cr - sy * sr, x],
        [sy * cp, sy * sp * sr + cy * cr, -sy * sp * cr + cy * sr, y],
        [sp, -cp * sr, cp * cr, z],
        [0.0, 0.0, 0.0, 1.0]])


def get_pairwise_transformation(base_data_dict, max_cav, proj_first):
    """``T[i, j] = T_j^-1 T_i`` for the agents in ``base_data_dict`` (insertion order), identity padded."""
    t = np.tile(np.eye(4), (max_cav, max_cav, 1, 1))
    if proj_first:
        return t
    world = [x_to_world(c['params']['lidar_pose']) for c in base_data_dict.values()]
    for i, ti in enumerate(world):
        for j, tj in enumerate(world):
            if i != j:
                t[i, j] = np.linalg.solve(tj, ti)
    return t
