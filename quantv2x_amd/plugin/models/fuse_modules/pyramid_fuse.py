"""``weighted_fuse`` of the HEAL Pyramid fusion (mirror of ``opencood/models/fuse_modules/pyramid_fuse.py:17-62``): the
occupancy-score-weighted fusion of one pyramid scale.  First piece of SURVEY.md §8(f) rank 3; on the GPU it is
``qv2x_pyramid_weighted_fuse_f32``.  The ``PyramidFusion`` module around it (ResNeXt multi-scale backbone) is not built yet."""
import torch
import torch.nn as nn

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


class PyramidFusion(nn.Module):
    """``PyramidFusion`` (``pyramid_fuse.py:64-179``): a ``ResNetBEVBackbone`` whose ``resnet`` is ResNeXt (grouped ``Bottleneck``s,
    32 x 4d, expansion 1) when ``resnext`` is set, plus one 1x1 occupancy head per level (``single_head_{i}``).  ``forward_collab``:
    multiscale features of every agent -> per level: occupancy map, ``score = sigmoid(occ) + 1e-4``, ``weighted_fuse`` -> deblocks."""

    def __init__(self, model_cfg, input_channels=64):
        from ..sub_modules.base_bev_backbone_resnet import ResNetBEVBackbone
        from ..sub_modules.resblock import Bottleneck, ResNetModified
        super().__init__()
        base = ResNetBEVBackbone(model_cfg, input_channels)
        self.model_cfg, self.num_levels, self.num_bev_features = base.model_cfg, base.num_levels, base.num_bev_features
        self.resnet, self.deblocks = base.resnet, base.deblocks
        self.stage = model_cfg["stage"]
        if model_cfg["resnext"]:
            Bottleneck.expansion = 1
            self.resnet = ResNetModified(Bottleneck, model_cfg['layer_nums'], model_cfg['layer_strides'], model_cfg['num_filters'],
                                         inplanes=model_cfg.get('inplanes', 64), groups=32, width_per_group=4)
        self.align_corners = model_cfg.get('align_corners', False)
        for i in range(self.num_levels):
            setattr(self, f"single_head_{i}", nn.Conv2d(model_cfg["num_filters"][i], 1, kernel_size=1))

    def get_multiscale_feature(self, x):
        return self.resnet(x)

    def decode_multiscale_feature(self, feats):
        ups = [self.deblocks[i](feats[i]) if len(self.deblocks) > 0 else feats[i] for i in range(self.num_levels)]
        x = torch.cat(ups, dim=1) if self.num_levels > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self.deblocks[-1](x)
        return x

    def forward_single(self, spatial_features):
        feats = self.get_multiscale_feature(spatial_features)
        occ = [getattr(self, f"single_head_{i}")(feats[i]) for i in range(self.num_levels)]
        return self.decode_multiscale_feature(feats), occ

    def forward_collab(self, spatial_features, record_len, affine_matrix, agent_modality_list=None, cam_crop_info=None):
        if cam_crop_info:
            raise NotImplementedError("camera crop masks: LiDAR modalities only on this path")
        feats = self.get_multiscale_feature(spatial_features)
        fused, occ = [], []
        for i in range(self.num_levels):
            o = getattr(self, f"single_head_{i}")(feats[i])
            occ.append(o)
            fused.append(weighted_fuse(feats[i], torch.sigmoid(o) + 1e-4, record_len, affine_matrix, self.align_corners))
        return self.decode_multiscale_feature(fused), occ

    def forward(self, spatial_features, record_len=None, affine_matrix=None, agent_modality_list=None, cam_crop_info=None):
        if self.stage == "single":
            return self.forward_single(spatial_features)
        if record_len is None or affine_matrix is None:
            raise ValueError("record_len and affine_matrix are required for forward_collab()")
        return self.forward_collab(spatial_features, record_len, affine_matrix, agent_modality_list, cam_crop_info)
