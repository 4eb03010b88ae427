"""Pillar feature network -- host-side mirror of ``opencood/models/sub_modules/pillar_vfe.py``.

Same constructor arguments, attribute names and ``state_dict`` keys
(``pfn_layers.{i}.linear.weight``, ``pfn_layers.{i}.norm.*``) as the reference
(PFNLayer ``pillar_vfe.py:10-53``, PillarVFE ``:56-155``).  This torch forward is the
PTQ *observer* surface (autograd-capable); the deployed int8 path is the HIP kernel
``qv2x_pfn_scatter_i8`` driven by ``quantv2x_amd.engine``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_ROWS_PER_GEMM = 50000  # the reference splits the pillar Linear at this many rows (pillar_vfe.py:29-39)


class PFNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        width = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, width, bias=not use_norm)
        if use_norm:
            self.norm = nn.BatchNorm1d(width, eps=1e-3, momentum=0.01)
        self.part = _ROWS_PER_GEMM

    def _project(self, pts):
        m = pts.shape[0]
        if m <= self.part:
            return self.linear(pts)
        pieces = [self.linear(pts[s:s + self.part]) for s in range(0, (m // self.part + 1) * self.part, self.part)]
        return torch.cat(pieces, dim=0)

    def forward(self, inputs):
        x = self._project(inputs)
        if self.use_norm:
            x = self.norm(x.transpose(1, 2)).transpose(1, 2)
        x = F.relu(x)
        pooled = x.max(dim=1, keepdim=True)[0]
        if self.last_vfe:
            return pooled
        return torch.cat([x, pooled.expand(-1, inputs.shape[1], -1)], dim=2)


class PillarVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range):
        super().__init__()
        self.model_cfg = model_cfg
        self.use_norm = model_cfg['use_norm']
        self.with_distance = model_cfg['with_distance']
        self.use_absolute_xyz = model_cfg['use_absolute_xyz']
        self.num_filters = model_cfg['num_filters']
        assert len(self.num_filters) > 0

        width_in = num_point_features + (6 if self.use_absolute_xyz else 3) + (1 if self.with_distance else 0)
        widths = [width_in] + list(self.num_filters)
        self.pfn_layers = nn.ModuleList(
            PFNLayer(widths[i], widths[i + 1], self.use_norm, last_layer=(i >= len(widths) - 2))
            for i in range(len(widths) - 1))

        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    @staticmethod
    def get_paddings_indicator(actual_num, max_num, axis=0):
        """True where slot index < number of real points (``[M, max_num]`` for axis=0)."""
        shape = [1] * (actual_num.dim() + 1)
        shape[axis + 1] = -1
        slots = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)
        return actual_num.unsqueeze(axis + 1).int() > slots

    def augment(self, voxel_features, voxel_num_points, coords):
        """4 raw features -> 10 decorated features, padded slots zeroed (pillar_vfe.py:119-149)."""
        xyz = voxel_features[:, :, :3]
        mean = xyz.sum(dim=1, keepdim=True) / voxel_num_points.type_as(voxel_features).view(-1, 1, 1)
        f_cluster = xyz - mean
        dt = voxel_features.dtype
        f_center = torch.zeros_like(xyz)
        f_center[:, :, 0] = xyz[:, :, 0] - (coords[:, 3].to(dt).unsqueeze(1) * self.voxel_x + self.x_offset)
        f_center[:, :, 1] = xyz[:, :, 1] - (coords[:, 2].to(dt).unsqueeze(1) * self.voxel_y + self.y_offset)
        f_center[:, :, 2] = xyz[:, :, 2] - (coords[:, 1].to(dt).unsqueeze(1) * self.voxel_z + self.z_offset)
        parts = [voxel_features if self.use_absolute_xyz else voxel_features[..., 3:], f_cluster, f_center]
        if self.with_distance:
            parts.append(torch.norm(xyz, 2, 2, keepdim=True))
        feats = torch.cat(parts, dim=-1)
        real = self.get_paddings_indicator(voxel_num_points, feats.shape[1], axis=0)
        return feats * real.unsqueeze(-1).type_as(voxel_features)

    def forward(self, batch_dict):
        feats = self.augment(batch_dict['voxel_features'], batch_dict['voxel_num_points'],
                             batch_dict['voxel_coords'])
        for pfn in self.pfn_layers:
            feats = pfn(feats)
        batch_dict['pillar_features'] = feats.squeeze()
        return batch_dict
