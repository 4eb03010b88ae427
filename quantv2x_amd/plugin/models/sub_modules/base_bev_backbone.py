"""Three-level BEV backbone; mirror of ``opencood/models/sub_modules/base_bev_backbone.py``.

``blocks.{i}`` = Sequential[ZeroPad2d(1), Conv3x3(stride, pad 0), BN, ReLU, (Conv3x3 p1, BN, ReLU) x layer_nums[i]]
``deblocks.{i}`` = Sequential[ConvTranspose2d(k = s = upsample_stride), BN, ReLU]      (reference ``:36-77``)
so the ``state_dict`` keys (``blocks.0.1.weight``, ``blocks.0.2.running_mean`` ...) are the reference's.
"""
import numpy as np
import torch
import torch.nn as nn


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = model_cfg.get('layer_nums', [])
        layer_strides = model_cfg.get('layer_strides', [])
        num_filters = model_cfg.get('num_filters', [])
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        up_strides = model_cfg.get('upsample_strides', [])
        up_filters = model_cfg.get('num_upsample_filter', [])
        assert len(up_strides) == len(up_filters)

        self.num_levels = len(layer_nums)
        widths_in = [input_channels] + list(num_filters[:-1])
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for lvl in range(self.num_levels):
            c = num_filters[lvl]
            seq = [nn.ZeroPad2d(1),
                   nn.Conv2d(widths_in[lvl], c, kernel_size=3, stride=layer_strides[lvl], padding=0, bias=False),
                   _bn(c), nn.ReLU()]
            for _ in range(layer_nums[lvl]):
                seq += [nn.Conv2d(c, c, kernel_size=3, padding=1, bias=False), _bn(c), nn.ReLU()]
            self.blocks.append(nn.Sequential(*seq))
            if len(up_strides) > 0:
                s = up_strides[lvl]
                if s >= 1:
                    up = nn.ConvTranspose2d(c, up_filters[lvl], s, stride=s, bias=False)
                else:  # fractional "upsample" = strided conv (reference :73 uses the removed np.int)
                    k = int(np.round(1 / s))
                    up = nn.Conv2d(c, up_filters[lvl], k, stride=k, bias=False)
                self.deblocks.append(nn.Sequential(up, _bn(up_filters[lvl]), nn.ReLU()))

        c_cat = sum(up_filters)
        if len(up_strides) > self.num_levels:
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_cat, c_cat, up_strides[-1], stride=up_strides[-1], bias=False),
                _bn(c_cat), nn.ReLU()))
        self.num_bev_features = c_cat

    def _merge(self, ups):
        x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self.deblocks[-1](x)
        return x

    def forward(self, x):
        ups = []
        for lvl in range(len(self.blocks)):
            x = self.blocks[lvl](x)
            ups.append(self.deblocks[lvl](x) if len(self.deblocks) > 0 else x)
        return self._merge(ups)

    def get_multiscale_feature(self, spatial_features):
        feats, x = [], spatial_features
        for blk in self.blocks:
            x = blk(x)
            feats.append(x)
        return feats

    def decode_multiscale_feature(self, x):
        ups = [self.deblocks[l](x[l]) if len(self.deblocks) > 0 else x[l] for l in range(self.num_levels)]
        return self._merge(ups)
