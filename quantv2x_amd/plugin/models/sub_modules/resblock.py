"""Residual blocks of the Pyramid (HEAL) model; mirror of ``opencood/models/sub_modules/resblock.py`` (``BasicBlock :20-66``,
``Bottleneck :69-128``, ``ResNetModified :131-227``): attribute names = the reference's ``state_dict`` keys
(``layer{i}.{b}.conv1/bn1/conv2/bn2[/conv3/bn3]/downsample.{0,1}``).

``ResNetModified`` returns the list of every level's output.  The width of a ``Bottleneck`` is
``planes * base_width / 64 * groups`` (ResNeXt 32 x 4d: twice ``planes``); the fusion backbone sets ``Bottleneck.expansion = 1``."""
from typing import List

import torch.nn as nn


def _conv3(cin, cout, stride=1, groups=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, groups=groups, bias=False)


def _conv1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1, self.bn1 = _conv3(inplanes, planes, stride), norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv3(planes, planes), norm_layer(planes)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + shortcut)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        if dilation > 1:
            raise NotImplementedError("dilated bottlenecks are outside the hot path")
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1, self.bn1 = _conv1(inplanes, width), norm_layer(width)
        self.conv2, self.bn2 = _conv3(width, width, stride, groups), norm_layer(width)
        self.conv3, self.bn3 = _conv1(width, planes * self.expansion), norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + shortcut)


class ResNetModified(nn.Module):
    def __init__(self, block, layers: List[int], layer_strides: List[int], num_filters: List[int], zero_init_residual=False,
                 groups=1, width_per_group=64, replace_stride_with_dilation=None, norm_layer=None, inplanes=64):
        super().__init__()
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self.block, self.layers, self.layer_strides, self.num_filters = block, layers, layer_strides, num_filters
        self.inplanes, self.dilation, self.groups, self.base_width = inplanes, 1, groups, width_per_group
        self.layernum = len(num_filters)
        for i in range(self.layernum):
            setattr(self, f"layer{i}", self._make_layer(block, num_filters[i], layers[i], stride=layer_strides[i]))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        shortcut = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            shortcut = nn.Sequential(_conv1(self.inplanes, planes * block.expansion, stride), self._norm_layer(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, shortcut, self.groups, self.base_width, self.dilation, self._norm_layer)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation, norm_layer=self._norm_layer)
                for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        feats = []
        for i in range(self.layernum):
            x = getattr(self, f"layer{i}")(x)
            feats.append(x)
        return feats
