"""Quantized wrappers of the hot-path blocks; mirror of the hot-path classes of
``opencood/quant/quant_block.py``: ``BaseQuantBlock :45-65``, ``QuantBaseBEVBackbone :243-335``,
``QuantDoubleConv / QuantDownsampleConv :552-586``, ``QuantPFNLayer :589-629`` (extra activation
quantizer after the ReLU, before the max), ``QuantPillarVFE :632-715``, ``QuantPointPillar :718-741``,
``QuantNaiveCompressor :1543-1570``; the Pyramid model's blocks ``QuantBasicBlock :68-97``, ``QuantBottleneck :100-131``,
``QuantResNetModified :338-395``, ``QuantResNetBEVBackbone :398-459``, ``QuantPyramidFusion :462-549``; registries ``opencood_specials :1581-1591``,
``specials_unquantized_names :1599-1615``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..models.fuse_modules.pyramid_fuse import PyramidFusion, weighted_fuse
from ..models.heter_encoders import SECOND, PointPillar
from ..models.sub_modules.base_bev_backbone import BaseBEVBackbone
from ..models.sub_modules.base_bev_backbone_resnet import ResNetBEVBackbone
from ..models.sub_modules.resblock import BasicBlock, Bottleneck, ResNetModified
from ..models.sub_modules.downsample_conv import DoubleConv, DownsampleConv
from ..models.sub_modules.naive_compress import NaiveCompressor
from ..models.sub_modules.pillar_vfe import PFNLayer, PillarVFE
from ..models.sub_modules.sparse_backbone_3d import VoxelBackBone8x
from ..models.sub_modules.sparse_ops import SparseConv3d, SparseConvTensor, SparseSequential, SubMConv3d
from .quant_layer import QuantModule, QuantSpconvModule, StraightThrough, UniformAffineQuantizer


class BaseQuantBlock(nn.Module):
    """A block whose inner ``QuantModule``s are switched together."""

    def __init__(self):
        super().__init__()
        self.use_weight_quant = False
        self.use_act_quant = False
        self.ignore_reconstruction = False
        self.trained = False

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, QuantModule):
                m.set_quant_state(weight_quant, act_quant)


def _wrap(conv, norm, act, wq, aq):
    q = QuantModule(conv, wq, aq)
    if norm is not None:
        q.norm_function = norm
    if act is not None:
        q.activation_function = act
    return q


class QuantBaseBEVBackbone(BaseQuantBlock):
    """``blocks.{i}`` = Sequential[ZeroPad2d, QuantModule, QuantModule, ...]; ``deblocks.{i}`` = Sequential[QuantModule]."""

    def __init__(self, basebevbackbone: BaseBEVBackbone, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        wq, aq = weight_quant_params, act_quant_params
        self.num_levels = basebevbackbone.num_levels
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for blk in basebevbackbone.blocks:
            seq = nn.Sequential(blk[0])                       # ZeroPad2d kept in place
            for at in range(1, len(blk), 3):                  # (conv, norm, relu) triplets
                seq.add_module(str(len(seq)), _wrap(blk[at], blk[at + 1], blk[at + 2], wq, aq))
            self.blocks.append(seq)
        for de in basebevbackbone.deblocks:
            self.deblocks.append(nn.Sequential(_wrap(de[0], de[1], de[2], wq, aq)))
        self.num_bev_features = basebevbackbone.num_bev_features

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
        return self._merge([self.deblocks[l](x[l]) if len(self.deblocks) > 0 else x[l]
                            for l in range(self.num_levels)])


class QuantDoubleConv(BaseQuantBlock):
    def __init__(self, double_conv: DoubleConv, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        dc = double_conv.double_conv
        self.double_conv = nn.Sequential(
            _wrap(dc[0], None, dc[1], weight_quant_params, act_quant_params),
            _wrap(dc[2], None, dc[3], weight_quant_params, act_quant_params))

    def forward(self, x):
        return self.double_conv[1](self.double_conv[0](x))


class QuantDownsampleConv(BaseQuantBlock):
    def __init__(self, downsample_conv: DownsampleConv, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        self.layers = nn.ModuleList(QuantDoubleConv(l, weight_quant_params, act_quant_params)
                                    for l in downsample_conv.layers)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class QuantPFNLayer(BaseQuantBlock):
    """Quantized pillar Linear; the ReLU is applied *outside* the QuantModule and is followed by a
    second activation quantizer (``self.act_quantizer``) before the max over points."""

    def __init__(self, pfn_layer: PFNLayer, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        self.last_vfe = pfn_layer.last_vfe
        self.use_norm = pfn_layer.use_norm
        self.part = pfn_layer.part
        self.linear = QuantModule(pfn_layer.linear, weight_quant_params, act_quant_params)
        if self.use_norm:
            self.linear.norm_function = pfn_layer.norm
        self.act_quantizer = UniformAffineQuantizer(**act_quant_params)

    def forward(self, inputs):
        m = inputs.shape[0]
        if m > self.part:
            x = torch.cat([self.linear(inputs[s:s + self.part])
                           for s in range(0, (m // self.part + 1) * self.part, self.part)], dim=0)
        else:
            x = self.linear(inputs)
        x = F.relu(x)
        if self.use_act_quant:
            x = self.act_quantizer(x)
        pooled = x.max(dim=1, keepdim=True)[0]
        if self.last_vfe:
            return pooled
        return torch.cat([x, pooled.repeat(1, inputs.shape[1], 1)], dim=2)


class QuantPillarVFE(nn.Module):
    def __init__(self, pillar_vfe: PillarVFE, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        for attr in ('with_distance', 'use_absolute_xyz', 'voxel_x', 'voxel_y', 'voxel_z',
                     'x_offset', 'y_offset', 'z_offset'):
            setattr(self, attr, getattr(pillar_vfe, attr))
        self.pfn_layers = nn.ModuleList(QuantPFNLayer(l, weight_quant_params, act_quant_params)
                                        for l in pillar_vfe.pfn_layers)

    get_paddings_indicator = staticmethod(PillarVFE.get_paddings_indicator)
    augment = PillarVFE.augment

    def forward(self, batch_dict):
        feats = self.augment(batch_dict['voxel_features'], batch_dict['voxel_num_points'],
                             batch_dict['voxel_coords'])
        for pfn in self.pfn_layers:
            feats = pfn(feats)
        batch_dict['pillar_features'] = feats.squeeze()
        return batch_dict


class QuantPointPillar(nn.Module):
    def __init__(self, point_pillar: PointPillar, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        self.pillar_vfe = QuantPillarVFE(point_pillar.pillar_vfe, weight_quant_params, act_quant_params)
        self.scatter = point_pillar.scatter

    def forward(self, data_dict, modality_name):
        src = data_dict[f'inputs_{modality_name}']
        batch = {k: src[k] for k in ('voxel_features', 'voxel_coords', 'voxel_num_points')}
        return self.scatter(self.pillar_vfe(batch))['spatial_features']


class QuantVoxelBackBone8x(BaseQuantBlock):
    """``VoxelBackBone8x`` with every sparse convolution wrapped (reference ``quant_block.py:988-1034``): a BatchNorm1d / ReLU that
    follows a convolution inside a ``SparseSequential`` becomes that wrapper's norm / activation."""

    def __init__(self, voxel_backbone: VoxelBackBone8x, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        self.model_cfg = voxel_backbone.model_cfg
        self.sparse_shape = voxel_backbone.sparse_shape
        self.num_point_features = voxel_backbone.num_point_features
        self.backbone_channels = voxel_backbone.backbone_channels

        def wrap(seq):
            out = SparseSequential()
            last = None
            for i, layer in enumerate(seq):
                if isinstance(layer, (SubMConv3d, SparseConv3d)):
                    last = QuantSpconvModule(layer, weight_quant_params, act_quant_params)
                    out.add_module(f"quant_conv_{i}", last)
                elif isinstance(layer, nn.BatchNorm1d) and last is not None:
                    last.norm_function = layer
                elif isinstance(layer, nn.ReLU) and last is not None:
                    last.activation_function = layer
                elif isinstance(layer, nn.Sequential):
                    out.add_module(f"layer_{i}", wrap(layer))
            return out

        for stage in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
            setattr(self, stage, wrap(getattr(voxel_backbone, stage)))

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant, self.use_act_quant = weight_quant, act_quant
        for m in self.modules():
            if isinstance(m, QuantSpconvModule):
                m.set_quant_state(weight_quant, act_quant)

    def forward(self, x):
        for stage in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
            x = getattr(self, stage)(x)
        return x


class QuantSECOND(nn.Module):
    """reference ``quant_block.py:1037-1078``: the mean VFE and the height compression stay as they are."""

    def __init__(self, second: SECOND, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        self.vfe = second.vfe
        self.map_to_bev = second.map_to_bev
        self.spconv_block = QuantVoxelBackBone8x(second.spconv_block, weight_quant_params, act_quant_params)

    def forward(self, data_dict, modality_name):
        src = data_dict[f'inputs_{modality_name}']
        batch = {k: src[k] for k in ('voxel_features', 'voxel_coords', 'voxel_num_points')}
        batch['batch_size'] = int(batch['voxel_coords'][:, 0].max().item()) + 1
        batch = self.vfe(batch)
        x = SparseConvTensor(features=batch['voxel_features'], indices=batch['voxel_coords'].int(),
                             spatial_shape=self.spconv_block.sparse_shape, batch_size=batch['batch_size'])
        batch['encoded_spconv_tensor'] = self.spconv_block(x)
        return self.map_to_bev(batch)['spatial_features']


class QuantNaiveCompressor(BaseQuantBlock):
    def __init__(self, naive_compressor: NaiveCompressor, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        e, d, wq, aq = naive_compressor.encoder, naive_compressor.decoder, weight_quant_params, act_quant_params
        self.encoder = nn.Sequential(_wrap(e[0], e[1], e[2], wq, aq))
        self.decoder = nn.Sequential(_wrap(d[0], d[1], d[2], wq, aq), _wrap(d[3], d[4], d[5], wq, aq))

    def forward(self, x):
        return self.decoder(self.encoder(x))

class _QuantResidual(BaseQuantBlock):
    """conv branch (last conv without output quantizer) + shortcut -> add in fp32 -> ReLU -> the block's own quantizer."""

    def _shortcut(self, block, wq, aq):
        if block.downsample is None:
            return None
        q = QuantModule(block.downsample[0], wq, aq, disable_act_quant=True)
        q.norm_function = block.downsample[1]
        return q

    def _finish(self, out, x):
        out += x if self.downsample is None else self.downsample(x)
        out = self.activation_function(out)
        return self.act_quantizer(out) if self.use_act_quant else out


class QuantBasicBlock(_QuantResidual):
    def __init__(self, basic_block: BasicBlock, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__()
        wq, aq = weight_quant_params, act_quant_params
        self.conv1 = _wrap(basic_block.conv1, basic_block.bn1, basic_block.relu, wq, aq)
        self.conv2 = QuantModule(basic_block.conv2, wq, aq, disable_act_quant=True)
        self.conv2.norm_function = basic_block.bn2
        self.downsample = self._shortcut(basic_block, wq, aq)
        self.activation_function = basic_block.relu
        self.act_quantizer = UniformAffineQuantizer(**aq)

    def forward(self, x):
        return self._finish(self.conv2(self.conv1(x)), x)


class QuantBottleneck(_QuantResidual):
    def __init__(self, bottleneck: Bottleneck, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__()
        wq, aq = weight_quant_params, act_quant_params
        self.conv1 = _wrap(bottleneck.conv1, bottleneck.bn1, bottleneck.relu, wq, aq)
        self.conv2 = _wrap(bottleneck.conv2, bottleneck.bn2, bottleneck.relu, wq, aq)
        self.conv3 = QuantModule(bottleneck.conv3, wq, aq, disable_act_quant=True)
        self.conv3.norm_function = bottleneck.bn3
        self.downsample = self._shortcut(bottleneck, wq, aq)
        self.activation_function = bottleneck.relu
        self.act_quantizer = UniformAffineQuantizer(**aq)

    def forward(self, x):
        return self._finish(self.conv3(self.conv2(self.conv1(x))), x)


class QuantResNetModified(BaseQuantBlock):
    _TWINS = {BasicBlock: QuantBasicBlock, Bottleneck: QuantBottleneck}

    def __init__(self, resnet_modified: ResNetModified, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        self._norm_layer, self.layernum, self.block = resnet_modified._norm_layer, resnet_modified.layernum, resnet_modified.block
        if self.block not in self._TWINS:
            raise ValueError(f"Unsupported block type: {self.block}. Please add it to block_mapping.")
        twin = self._TWINS[self.block]
        for i in range(self.layernum):
            seq = [twin(b, weight_quant_params, act_quant_params) if isinstance(b, self.block) else b
                   for b in getattr(resnet_modified, f"layer{i}")]
            setattr(self, f"layer{i}", nn.Sequential(*seq))

    def forward(self, x):
        feats = []
        for i in range(self.layernum):
            x = getattr(self, f"layer{i}")(x)
            feats.append(x)
        return feats


class QuantResNetBEVBackbone(BaseQuantBlock):
    def __init__(self, resnet_bev_backbone: ResNetBEVBackbone, weight_quant_params={}, act_quant_params={}):
        super().__init__()
        src = resnet_bev_backbone
        self.model_cfg, self.num_levels, self.num_bev_features = src.model_cfg, src.num_levels, src.num_bev_features
        self.resnet = QuantResNetModified(src.resnet, weight_quant_params, act_quant_params)
        self.deblocks = nn.ModuleList(nn.Sequential(_wrap(de[0], de[1], de[2], weight_quant_params, act_quant_params))
                                      for de in src.deblocks)

    def get_multiscale_feature(self, spatial_features):
        return self.resnet(spatial_features)

    def decode_multiscale_feature(self, x):
        ups = [self.deblocks[i](x[i]) if len(self.deblocks) > 0 else x[i] for i in range(self.num_levels)]
        x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self.deblocks[-1](x)
        return x

    def forward(self, spatial_features):
        return self.decode_multiscale_feature(self.resnet(spatial_features))

    def get_layer_i_feature(self, spatial_features, layer_i):
        return getattr(self.resnet, f"layer{layer_i}")(spatial_features)


class QuantPyramidFusion(QuantResNetBEVBackbone):
    """The occupancy heads are plain ``QuantModule``s (their own output quantizer stays on); the fused maps that feed the
    deblocks are fp32 (``weighted_fuse`` of fake-quantized features with softmax weights)."""

    def __init__(self, pyramid_fusion: PyramidFusion, weight_quant_params={}, act_quant_params={}):
        super().__init__(pyramid_fusion, weight_quant_params, act_quant_params)
        self.stage, self.align_corners = pyramid_fusion.stage, pyramid_fusion.align_corners
        for i in range(self.num_levels):
            setattr(self, f"single_head_{i}", QuantModule(getattr(pyramid_fusion, f"single_head_{i}"), weight_quant_params, act_quant_params))

    def forward_single(self, spatial_features):
        feats = self.get_multiscale_feature(spatial_features)
        occ = [getattr(self, f"single_head_{i}")(feats[i]) for i in range(self.num_levels)]
        return self.decode_multiscale_feature(feats), occ

    def forward_collab(self, spatial_features, record_len, affine_matrix, agent_modality_list=None, cam_crop_info=None):
        if cam_crop_info:
            raise NotImplementedError("camera crop masks: LiDAR modalities only on this path")
        feats = self.get_multiscale_feature(spatial_features)
        fused, occ = [], []
        for i in range(self.num_levels):
            o = getattr(self, f"single_head_{i}")(feats[i])
            occ.append(o)
            fused.append(weighted_fuse(feats[i], torch.sigmoid(o) + 1e-4, record_len, affine_matrix, self.align_corners))
        return self.decode_multiscale_feature(fused), occ

    def forward(self, spatial_features, record_len=None, affine_matrix=None, agent_modality_list=None, cam_crop_info=None):
        if self.stage == "single":
            return self.forward_single(spatial_features)
        if record_len is None or affine_matrix is None:
            raise ValueError("record_len and affine_matrix are required for forward_collab()")
        return self.forward_collab(spatial_features, record_len, affine_matrix, agent_modality_list, cam_crop_info)


specials = {BasicBlock: QuantBasicBlock, Bottleneck: QuantBottleneck}

opencood_specials = {
    PyramidFusion: QuantPyramidFusion,
    ResNetBEVBackbone: QuantResNetBEVBackbone,
    BaseBEVBackbone: QuantBaseBEVBackbone,
    DownsampleConv: QuantDownsampleConv,
    PointPillar: QuantPointPillar,
    SECOND: QuantSECOND,
    NaiveCompressor: QuantNaiveCompressor,
}

specials_unquantized = []

# children with these *names* are neither folded nor quantized (the codebook stays fp32)
specials_unquantized_names = ['aligner_m1', 'aligner_m2', 'codebook']
