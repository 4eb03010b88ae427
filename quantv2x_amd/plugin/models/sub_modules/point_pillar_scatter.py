"""Pillar -> dense BEV canvas; mirror of ``opencood/models/sub_modules/point_pillar_scatter.py:19-75``.

Canvas index is ``z + y * nx + x`` (reference ``:57``); output ``[B, C*nz, ny, nx]``.
"""
import torch
import torch.nn as nn


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg['num_features']
        self.nx, self.ny, self.nz = (int(v) for v in model_cfg['grid_size'])
        assert self.nz == 1

    def forward(self, batch_dict):
        pillars, coords = batch_dict['pillar_features'], batch_dict['voxel_coords']
        n_batch = int(coords[:, 0].max().item()) + 1
        canvas = torch.zeros(n_batch, self.nz * self.ny * self.nx, self.num_bev_features,
                             dtype=pillars.dtype, device=pillars.device)
        cell = (coords[:, 1] + coords[:, 2] * self.nx + coords[:, 3]).long()
        canvas[coords[:, 0].long(), cell] = pillars
        batch_dict['spatial_features'] = canvas.transpose(1, 2).contiguous().view(
            n_batch, self.num_bev_features * self.nz, self.ny, self.nx)
        return batch_dict
