"""Sparse 3-D convolution modules with the spconv 2.x surface the SECOND encoder uses (``SparseConvTensor``, ``SubMConv3d``,
``SparseConv3d``, ``SparseSequential``), written in plain torch.

spconv is a third-party wheel that is absent from the reference tree and from this image (parity
unpinned), so this module restates its published semantics instead of binding it:

* a sparse tensor is ``features [N, C]`` + ``indices [N, 4] = (batch, z, y, x)`` over ``spatial_shape = (D, H, W)``; sites that are
  absent hold the real value 0;
* ``SparseConv3d``: a cross-correlation like ``F.conv3d`` on the densified input, evaluated only at output positions whose window
  holds at least one active input (those become the active output sites);
* ``SubMConv3d``: stride 1, "same" geometry, evaluated only AT the input's active sites (the active set does not grow);
* weights are ``[C_out, kz, ky, kx, C_in]`` (spconv 2.x "KRSC"), so a ``channel_wise`` weight quantizer scales per output channel.

Output rows of a ``SparseConv3d`` are in raster order of (batch, z, y, x); spconv's own order is hash-table order and nothing
downstream depends on it (``HeightCompression`` densifies).
"""
import math
from typing import Sequence

import torch
import torch.nn as nn


def _triple(v):
    return tuple(int(t) for t in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}

    def replace_feature(self, features):
        out = SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size)
        out.indice_dict = self.indice_dict
        return out

    def keys(self, indices=None, shape=None):
        i = (self.indices if indices is None else indices).long()
        d, h, w = self.spatial_shape if shape is None else shape
        return ((i[:, 0] * d + i[:, 1]) * h + i[:, 2]) * w + i[:, 3]

    def dense(self):
        d, h, w = self.spatial_shape
        c = self.features.shape[1]
        out = self.features.new_zeros(self.batch_size * d * h * w, c)
        out[self.keys()] = self.features
        return out.view(self.batch_size, d, h, w, c).permute(0, 4, 1, 2, 3).contiguous()


class _Lookup:
    """row of a site by coordinate: sorted linear keys + searchsorted (no dense index volume on the host)."""

    def __init__(self, keys):
        self.sorted, self.perm = torch.sort(keys)

    def __call__(self, q, ok):
        if self.sorted.numel() == 0:
            return torch.zeros_like(q), torch.zeros_like(ok)
        pos = torch.searchsorted(self.sorted, q).clamp_(max=self.sorted.numel() - 1)
        hit = ok & (self.sorted[pos] == q)
        return self.perm[pos], hit


class _SparseConvBase(nn.Module):
    subm = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        if self.subm:
            self.stride, self.padding = (1, 1, 1), tuple(k // 2 for k in self.kernel_size)
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=math.sqrt(5))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def out_shape(self, shape: Sequence[int]):
        if self.subm:
            return list(shape)
        return [(n + 2 * p - k) // s + 1 for n, p, k, s in zip(shape, self.padding, self.kernel_size, self.stride)]

    def _out_sites(self, x: SparseConvTensor, oshape):
        """Active outputs of a strided sparse convolution: every position whose window holds an active input."""
        i = x.indices.long()
        cand = []
        for kz in range(self.kernel_size[0]):
            for ky in range(self.kernel_size[1]):
                for kx in range(self.kernel_size[2]):
                    num = [i[:, 1 + a] + self.padding[a] - k for a, k in enumerate((kz, ky, kx))]
                    ok = torch.ones_like(num[0], dtype=torch.bool)
                    o = []
                    for a in range(3):
                        ok &= (num[a] % self.stride[a] == 0)
                        q = torch.div(num[a], self.stride[a], rounding_mode='floor')
                        ok &= (q >= 0) & (q < oshape[a])
                        o.append(q)
                    cand.append((((i[:, 0] * oshape[0] + o[0]) * oshape[1] + o[1]) * oshape[2] + o[2])[ok])
        keys = torch.unique(torch.cat(cand))                 # sorted: raster order
        w = keys % oshape[2]
        h = (keys // oshape[2]) % oshape[1]
        d = (keys // (oshape[2] * oshape[1])) % oshape[0]
        b = keys // (oshape[2] * oshape[1] * oshape[0])
        return torch.stack([b, d, h, w], 1).to(x.indices.dtype)

    def forward(self, x: SparseConvTensor, weight=None, bias=None):
        weight = self.weight if weight is None else weight
        bias = self.bias if bias is None else bias
        ishape = x.spatial_shape
        oshape = self.out_shape(ishape)
        oidx = x.indices if self.subm else self._out_sites(x, oshape)
        look = _Lookup(x.keys())
        o = oidx.long()
        out = x.features.new_zeros(o.shape[0], self.out_channels)
        for kz in range(self.kernel_size[0]):
            for ky in range(self.kernel_size[1]):
                for kx in range(self.kernel_size[2]):
                    src = [o[:, 1 + a] * self.stride[a] - self.padding[a] + k for a, k in enumerate((kz, ky, kx))]
                    ok = torch.ones_like(src[0], dtype=torch.bool)
                    for a in range(3):
                        ok &= (src[a] >= 0) & (src[a] < ishape[a])
                    q = ((o[:, 0] * ishape[0] + src[0]) * ishape[1] + src[1]) * ishape[2] + src[2]
                    rows, hit = look(q, ok)
                    if hit.any():
                        out[hit] += x.features[rows[hit]] @ weight[:, kz, ky, kx, :].t()
        if bias is not None:
            out = out + bias
        res = SparseConvTensor(out, oidx, oshape, x.batch_size)
        res.indice_dict = x.indice_dict
        return res


class SparseConv3d(_SparseConvBase):
    subm = False


class SubMConv3d(_SparseConvBase):
    subm = True


def is_sparse_module(m) -> bool:
    return isinstance(m, (_SparseConvBase, SparseSequential)) or getattr(m, "is_sparse_conv", False)


class SparseSequential(nn.Sequential):
    """Sparse modules see the tensor; anything else (BatchNorm1d, ReLU) sees ``.features``."""

    def forward(self, input):
        for m in self:
            if is_sparse_module(m) or isinstance(input, torch.Tensor):
                input = m(input)
            else:
                input = input.replace_feature(m(input.features))
        return input
