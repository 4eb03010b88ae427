"""ResNet BEV backbone of the Pyramid model; mirror of ``opencood/models/sub_modules/base_bev_backbone_resnet.py:13-137``:
``resnet`` (``ResNetModified`` of ``BasicBlock``s) + per-level ``deblocks`` (``ConvTranspose2d(k = s)`` + BN(eps 1e-3) + ReLU), concat.
``get_multiscale_feature`` / ``decode_multiscale_feature`` split the forward for the multiscale fusion."""
import torch
import torch.nn as nn

from .resblock import BasicBlock, ResNetModified


class ResNetBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels=64):
        super().__init__()
        self.model_cfg = model_cfg
        nums, strides, filters = (model_cfg.get(k, []) for k in ('layer_nums', 'layer_strides', 'num_filters'))
        assert len(nums) == len(strides) == len(filters)
        ups, up_filters = model_cfg.get('upsample_strides', []), model_cfg.get('num_upsample_filter', [])
        assert len(ups) == len(up_filters)
        self.resnet = ResNetModified(BasicBlock, nums, strides, filters, inplanes=model_cfg.get('inplanes', 64))
        self.num_levels = len(nums)
        self.deblocks = nn.ModuleList()
        for idx in range(self.num_levels):
            if len(ups) == 0:
                break
            if ups[idx] < 1:
                raise NotImplementedError("fractional upsample strides (strided conv deblocks) are outside the hot path")
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(filters[idx], up_filters[idx], ups[idx], stride=ups[idx], bias=False),
                nn.BatchNorm2d(up_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
        c_in = sum(up_filters)
        if len(ups) > self.num_levels:
            self.deblocks.append(nn.Sequential(nn.ConvTranspose2d(c_in, c_in, ups[-1], stride=ups[-1], bias=False),
                                               nn.BatchNorm2d(c_in, eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features = c_in

    def get_multiscale_feature(self, spatial_features):
        return self.resnet(spatial_features)

    def decode_multiscale_feature(self, feats):
        ups = [self.deblocks[i](feats[i]) if len(self.deblocks) > 0 else feats[i] for i in range(self.num_levels)]
        x = torch.cat(ups, dim=1) if self.num_levels > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self.deblocks[-1](x)
        return x

    def forward(self, spatial_features):
        return self.decode_multiscale_feature(self.resnet(spatial_features))

    def get_layer_i_feature(self, spatial_features, layer_i):
        return getattr(self.resnet, f"layer{layer_i}")(spatial_features)
