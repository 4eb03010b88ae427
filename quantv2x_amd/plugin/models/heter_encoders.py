"""Encoder plugins looked up by ``core_method`` (reference ``heter_model_baseline.py:47-59``).

The LiDAR encoders: PointPillar (the hot path, reference ``opencood/models/heter_encoders.py:22-50``) and SECOND
(reference ``:52-81``; its sparse convolutions come from ``sub_modules/sparse_ops`` because spconv is not
available).  LiftSplatShoot (camera) is out of scope.
"""
import numpy as np
import torch.nn as nn

from .sub_modules.height_compression import HeightCompression
from .sub_modules.mean_vfe import MeanVFE
from .sub_modules.pillar_vfe import PillarVFE
from .sub_modules.point_pillar_scatter import PointPillarScatter
from .sub_modules.sparse_backbone_3d import VoxelBackBone8x


class PointPillar(nn.Module):
    def __init__(self, args):
        super().__init__()
        extent = np.array(args['lidar_range'][3:6]) - np.array(args['lidar_range'][0:3])
        # like the reference, the grid size is written back into the caller's config
        args['point_pillar_scatter']['grid_size'] = np.round(extent / np.array(args['voxel_size'])).astype(np.int64)
        self.pillar_vfe = PillarVFE(args['pillar_vfe'], num_point_features=4,
                                    voxel_size=args['voxel_size'], point_cloud_range=args['lidar_range'])
        self.scatter = PointPillarScatter(args['point_pillar_scatter'])

    def forward(self, data_dict, modality_name):
        src = data_dict[f'inputs_{modality_name}']
        batch = {k: src[k] for k in ('voxel_features', 'voxel_coords', 'voxel_num_points')}
        return self.scatter(self.pillar_vfe(batch))['spatial_features']


class SECOND(nn.Module):
    def __init__(self, args):
        super().__init__()
        rng = np.array(args['lidar_range'])
        grid_size = np.round((rng[3:6] - rng[:3]) / np.array(args['voxel_size'])).astype(np.int64)
        self.vfe = MeanVFE(args['mean_vfe'], args['mean_vfe']['num_point_features'])
        self.spconv_block = VoxelBackBone8x(args['spconv'], input_channels=args['spconv']['num_features_in'], grid_size=grid_size)
        self.map_to_bev = HeightCompression(args['map2bev'])

    def forward(self, data_dict, modality_name):
        src = data_dict[f'inputs_{modality_name}']
        batch = {k: src[k] for k in ('voxel_features', 'voxel_coords', 'voxel_num_points')}
        batch['batch_size'] = int(batch['voxel_coords'][:, 0].max()) + 1
        return self.map_to_bev(self.spconv_block(self.vfe(batch)))['spatial_features']
