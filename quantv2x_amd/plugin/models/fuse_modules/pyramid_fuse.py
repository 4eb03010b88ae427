"""``weighted_fuse`` of the HEAL Pyramid fusion (mirror of ``opencood/models/fuse_modules/pyramid_fuse.py:17-62``): the
occupancy-score-weighted fusion of one pyramid scale.  First piece of SURVEY.md §8(f) rank 3; on the GPU it is
``qv2x_pyramid_weighted_fuse_f32``.  The ``PyramidFusion`` module around it (ResNeXt multi-scale backbone) is not built yet."""
import torch

from ..sub_modules.torch_transformation_utils import warp_affine_simple
from .fusion_in_one import regroup


def weighted_fuse(x, score, record_len, affine_matrix, align_corners):
    """x [sum(n_cav), C, H, W], score [sum(n_cav), 1, H, W], affine_matrix [B, L, L, 2, 3] -> [B, C, H, W]"""
    _, _, h, w = x.shape
    feats, scores = regroup(x, record_len), regroup(score, record_len)
    fused = []
    for b in range(affine_matrix.shape[0]):
        n = record_len[b]
        t = affine_matrix[b][:n, :n][0]                                  # every agent into the ego (agent 0) frame
        f_ego = warp_affine_simple(feats[b], t, (h, w), align_corners=align_corners)
        s_ego = warp_affine_simple(scores[b], t, (h, w), align_corners=align_corners)
        s_ego = s_ego.masked_fill(s_ego == 0, -float('inf'))
        p = torch.softmax(s_ego, dim=0)
        p = torch.where(torch.isnan(p), torch.zeros_like(p), p)
        fused.append(torch.sum(f_ego * p, dim=0))
    return torch.stack(fused)
