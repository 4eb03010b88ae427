"""Encoder plugins looked up by ``core_method`` (reference ``heter_model_baseline.py:47-59``).

Only the LiDAR PointPillar encoder of the hot path is provided (reference
``opencood/models/heter_encoders.py:22-50``); SECOND / LiftSplatShoot are out of scope.
"""
import numpy as np
import torch.nn as nn

from .sub_modules.pillar_vfe import PillarVFE
from .sub_modules.point_pillar_scatter import PointPillarScatter


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
