"""HEAL Pyramid-fusion host model (multi-class heads); mirror of ``opencood/models/heter_pyramid_collab_mc.py`` (ctor ``:23-133``,
forward ``:164-249``): per-modality encoder + ``ResNetBEVBackbone`` + ``AlignNet`` -> [codebook / compressor hook] ->
``PyramidFusion`` (ResNeXt levels, occupancy-weighted fusion, deblocks) -> ``shrink_conv`` -> shared 1x1 heads.

Same ``args`` schema, attribute names (``encoder_m1``, ``backbone_m1``, ``aligner_m1``, ``pyramid_backbone``, ``shrink_conv``,
``cls_head`` ...), output dict and ``state_dict`` keys as the reference.  LiDAR modalities only."""
import importlib
from collections import Counter, OrderedDict

import numpy as np
import torch
import torch.nn as nn

from ..utils.transformation_utils import normalize_pairwise_tfm
from .fuse_modules.pyramid_fuse import PyramidFusion
from .heter_model_baseline import find_class
from .sub_modules.base_bev_backbone_resnet import ResNetBEVBackbone
from .sub_modules.downsample_conv import DownsampleConv
from .sub_modules.feature_alignnet import AlignNet
from .sub_modules.naive_compress import NaiveCompressor


def modality_name(x) -> str:
    """'m1' from 'm1' / 1 / tensor(1) (``heter_pyramid_collab_mc.py:138-151``)."""
    if isinstance(x, str):
        return x
    if isinstance(x, torch.Tensor):
        return f"m{int(x.item())}"
    if isinstance(x, (int, np.integer)):
        return f"m{int(x)}"
    raise TypeError(f"Unexpected type for modality: {type(x)}")


class HeterPyramidCollabMC(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.modality_name_list = [k for k in args.keys() if k.startswith("m") and k[1:].isdigit()]
        self.num_class = args["num_class"]
        self.cav_range = args['lidar_range']
        self.sensor_type_dict = OrderedDict()
        self.cam_crop_info = {}
        encoders = importlib.import_module(__package__ + ".heter_encoders")
        for name in self.modality_name_list:
            cfg = args[name]
            self.sensor_type_dict[name] = cfg['sensor_type']
            if cfg['sensor_type'] != 'lidar':
                raise NotImplementedError("only LiDAR modalities are on the accelerated path (SURVEY.md §2)")
            enc_cls = find_class(encoders, cfg['core_method'])
            if enc_cls is None:
                raise NotImplementedError(f"encoder {cfg['core_method']!r} is outside the hot path")
            setattr(self, f"encoder_{name}", enc_cls(cfg['encoder_args']))
            setattr(self, f"depth_supervision_{name}", bool(cfg['encoder_args'].get("depth_supervision", False)))
            setattr(self, f"backbone_{name}", ResNetBEVBackbone(cfg['backbone_args']))
            setattr(self, f"aligner_{name}", AlignNet(cfg['aligner_args']))

        self.H = self.cav_range[4] - self.cav_range[1]
        self.W = self.cav_range[3] - self.cav_range[0]
        self.fake_voxel_size = 1

        self.compress = 'compressor' in args
        if self.compress:
            self.compressor = NaiveCompressor(args['compressor']['input_dim'], args['compressor']['compress_ratio'])

        fusion_args = args['fusion_backbone']
        if fusion_args.get("proj_first", False):
            raise NotImplementedError("proj_first (the ONNX-export fusion) is outside the hot path")
        self.pyramid_backbone = PyramidFusion(fusion_args)

        self.shrink_flag = 'shrink_header' in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args['shrink_header'])

        a, c = args['anchor_number'], args['num_class']
        self.cls_head = nn.Conv2d(args['in_head'], a * c * c, kernel_size=1)
        self.reg_head = nn.Conv2d(args['in_head'], 7 * a * c, kernel_size=1)
        self.dir_head = nn.Conv2d(args['in_head'], args['dir_args']['num_bins'] * a * c, kernel_size=1)
        self.model_train_init()

    def model_train_init(self):
        if self.compress:
            self.eval()
            for p in self.parameters():
                p.requires_grad_(False)
            self.compressor.train()
            for p in self.compressor.parameters():
                p.requires_grad_(True)

    def get_memory_footprint(self):
        total = sum(t.nelement() * t.element_size() for t in list(self.parameters()) + list(self.buffers()))
        return f"Model Memory Footprint: {total / (1024 ** 2):.2f} MB"

    # ---- forward in stages (the codebook subclasses and the deployed engine reuse them) ----
    def encode_agents(self, data_dict):
        """Per-agent BEV features in ``agent_modality_list`` order: [sum_N, C, H, W] (2x downsampled)."""
        agents = [modality_name(x) for x in data_dict['agent_modality_list']]
        present = Counter(agents)
        per_mod = {}
        for name in self.modality_name_list:
            if name in present:
                f = getattr(self, f"encoder_{name}")(data_dict, name)
                f = getattr(self, f"backbone_{name}")(f)
                per_mod[name] = getattr(self, f"aligner_{name}")(f)
        taken = {name: 0 for name in self.modality_name_list}
        rows = []
        for name in agents:
            rows.append(per_mod[name][taken[name]])
            taken[name] += 1
        return torch.stack(rows)

    def transform_features(self, feats, output_dict):
        return self.compressor(feats) if self.compress else feats

    def fuse_and_detect(self, feats, record_len, affine, agent_modality_list, output_dict):
        fused, occ = self.pyramid_backbone(feats, record_len, affine, agent_modality_list, self.cam_crop_info)
        if self.shrink_flag:
            fused = self.shrink_conv(fused)
        cls, reg, dr = self.cls_head(fused), self.reg_head(fused), self.dir_head(fused)
        output_dict.update({'cls_preds': cls, 'reg_preds': reg, 'dir_preds': dr, 'occ_single_list': occ,
                            'preds_tensor': torch.cat([cls, reg, dr], dim=1)})
        return output_dict

    def forward(self, data_dict):
        out = {'pyramid': 'collab'}
        affine = normalize_pairwise_tfm(data_dict['pairwise_t_matrix'], self.H, self.W, self.fake_voxel_size)
        feats = self.transform_features(self.encode_agents(data_dict), out)
        return self.fuse_and_detect(feats, data_dict['record_len'], affine, data_dict['agent_modality_list'], out)
