"""Sparse volume -> BEV map; mirror of ``opencood/models/sub_modules/height_compression.py:4-27``."""
import torch.nn as nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg['feature_num']

    def forward(self, batch_dict):
        vol = batch_dict['encoded_spconv_tensor'].dense()             # [N, C, D, H, W]
        n, c, d, h, w = vol.shape
        batch_dict['spatial_features'] = vol.view(n, c * d, h, w)    # channel = c * D + d
        batch_dict['spatial_features_stride'] = batch_dict.get('encoded_spconv_tensor_stride')
        return batch_dict
