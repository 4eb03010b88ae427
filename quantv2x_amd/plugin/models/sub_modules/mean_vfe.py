"""Mean voxel feature encoder of the SECOND encoder; mirror of ``opencood/models/sub_modules/mean_vfe.py:4-32``."""
import torch
import torch.nn as nn


class MeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        """``voxel_features [M, T, C]`` (unused slots zero) and ``voxel_num_points [M]`` -> the per-voxel mean ``[M, C]``."""
        total = batch_dict['voxel_features'].sum(dim=1)
        count = torch.clamp_min(batch_dict['voxel_num_points'].view(-1, 1), min=1.0).type_as(total)
        batch_dict['voxel_features'] = (total / count).contiguous()
        return batch_dict
