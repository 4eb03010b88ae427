"""Fusion operators with the reference's plugin signature
``fusion_net(x[sum_N, C, H, W], record_len[B], affine[B, L, L, 2, 3]) -> [B, C, H, W]``
(``opencood/models/fuse_modules/fusion_in_one.py``: ScaledDotProductAttention ``:14-45``,
regroup ``:48-51``, MaxFusion ``:87-124``, AttFusion ``:126-151``)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..sub_modules.torch_transformation_utils import warp_affine_simple


class ScaledDotProductAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.sqrt_dim = np.sqrt(dim)

    def forward(self, query, key, value):
        weights = F.softmax(torch.bmm(query, key.transpose(1, 2)) / self.sqrt_dim, -1)
        return torch.bmm(weights, value)


def regroup(x, record_len):
    ends = torch.cumsum(record_len, dim=0)
    return torch.tensor_split(x, ends[:-1].cpu())


def _warp_to_ego(x, record_len, affine_matrix):
    """Yield, per batch sample, all of its agents resampled into the ego (agent 0) frame."""
    h, w = x.shape[-2:]
    for b, feats in enumerate(regroup(x, record_len)):
        n = int(record_len[b])
        yield warp_affine_simple(feats, affine_matrix[b][:n, :n][0], (h, w))


class MaxFusion(nn.Module):
    def forward(self, x, record_len, affine_matrix):
        return torch.stack([w.max(dim=0)[0] for w in _warp_to_ego(x, record_len, affine_matrix)])


class AttFusion(nn.Module):
    """Per BEV cell, scaled-dot-product attention across agents with Q = K = V; the ego row is kept."""

    def __init__(self, feature_dims):
        super().__init__()
        self.att = ScaledDotProductAttention(feature_dims)

    def forward(self, xx, record_len, affine_matrix):
        c, h, w = xx.shape[1:]
        fused = []
        for warped in _warp_to_ego(xx, record_len, affine_matrix):
            n = warped.shape[0]
            tokens = warped.view(n, c, -1).permute(2, 0, 1)          # [H*W, n, C]
            ctx = self.att(tokens, tokens, tokens)
            fused.append(ctx.permute(1, 2, 0).view(n, c, h, w)[0])
        return torch.stack(fused)
