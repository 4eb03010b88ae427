"""``VoxelBackBone8x``: the 12-layer sparse 3-D convolution stack of the SECOND encoder (4 -> 16 -> 32 -> 64 -> 64 -> 128, 8x down
in x / y); mirror of ``opencood/models/sub_modules/sparse_backbone_3d.py:33-153`` over ``sparse_ops`` (spconv is not available).
"""
from functools import partial

import torch.nn as nn

from .sparse_ops import SparseConv3d, SparseConvTensor, SparseSequential, SubMConv3d


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm', norm_fn=None):
    if conv_type == 'subm':
        conv = SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False, indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


# (stage, [(c_in, c_out, conv_type, stride, padding, indice_key)])
_STAGES = (
    ("conv1", [(16, 16, 'subm', 1, 1, 'subm1')]),
    ("conv2", [(16, 32, 'spconv', 2, 1, 'spconv2'), (32, 32, 'subm', 1, 1, 'subm2'), (32, 32, 'subm', 1, 1, 'subm2')]),
    ("conv3", [(32, 64, 'spconv', 2, 1, 'spconv3'), (64, 64, 'subm', 1, 1, 'subm3'), (64, 64, 'subm', 1, 1, 'subm3')]),
    ("conv4", [(64, 64, 'spconv', 2, (0, 1, 1), 'spconv4'), (64, 64, 'subm', 1, 1, 'subm4'), (64, 64, 'subm', 1, 1, 'subm4')]),
)


class VoxelBackBone8x(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = grid_size[::-1] + [1, 0, 0]               # (z + 1, y, x)
        self.conv_input = SparseSequential(SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'),
                                           norm_fn(16), nn.ReLU())
        for stage, layers in _STAGES:
            setattr(self, stage, SparseSequential(*[
                post_act_block(ci, co, 3, norm_fn=norm_fn, stride=s, padding=p, indice_key=key, conv_type=kind)
                for ci, co, kind, s, p, key in layers]))
        self.num_point_features = self.model_cfg.get('num_features_out', 128)
        self.conv_out = SparseSequential(
            SparseConv3d(64, self.num_point_features, (3, 1, 1), stride=(2, 1, 1), padding=0, bias=False, indice_key='spconv_down2'),
            norm_fn(self.num_point_features), nn.ReLU())
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}

    def forward(self, batch_dict):
        x = SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                             spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        scales = {}
        for i, (stage, _) in enumerate(_STAGES):
            x = getattr(self, stage)(x)
            scales[f'x_conv{i + 1}'] = x
        batch_dict.update({'encoded_spconv_tensor': self.conv_out(x), 'encoded_spconv_tensor_stride': 8,
                           'multi_scale_3d_features': scales,
                           'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict
