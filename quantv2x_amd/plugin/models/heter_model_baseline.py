"""Host model of the hot path: per-modality encoder + BEV backbone + shrinker, selectable
fusion, shared 1x1 heads.  Mirror of ``opencood/models/heter_model_baseline.py`` (ctor ``:28-145``,
forward ``:169-265``): same ``args`` schema, attribute names (``encoder_m1``, ``backbone_m1``,
``shrinker_m1``, ``fusion_net``, ``cls_head`` ...), output dict and ``state_dict`` keys.
"""
import importlib
from collections import Counter, OrderedDict

import torch
import torch.nn as nn

from ..utils.transformation_utils import normalize_pairwise_tfm
from .fuse_modules.fusion_in_one import AttFusion, MaxFusion
from .sub_modules.base_bev_backbone import BaseBEVBackbone
from .sub_modules.downsample_conv import DownsampleConv
from .sub_modules.naive_compress import NaiveCompressor

_FUSIONS = {"max": lambda a: MaxFusion(), "att": lambda a: AttFusion(a['att']['feat_dim'])}


def find_class(module, wanted: str):
    """The reference's plugin lookup: case-insensitive class name with '_' removed (train_utils.py:272-291)."""
    wanted = wanted.replace('_', '').lower()
    hit = None
    for name, obj in module.__dict__.items():
        if name.lower() == wanted:
            hit = obj
    return hit


class HeterModelBaseline(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.fusion_method = args['fusion_method']
        self.modality_name_list = [k for k in args.keys() if k.startswith("m") and k[1:].isdigit()]
        self.ego_modality = args['ego_modality']
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
            setattr(self, f"backbone_{name}",
                    BaseBEVBackbone(cfg['backbone_args'], cfg['backbone_args'].get('inplanes', 64)))
            setattr(self, f"shrinker_{name}", DownsampleConv(cfg['shrink_header']))

        # metres covered by the feature map; used to normalise the pairwise transforms
        self.H = self.cav_range[4] - self.cav_range[1]
        self.W = self.cav_range[3] - self.cav_range[0]
        self.fake_voxel_size = 1

        self.supervise_single = bool(args.get("supervise_single", False))
        if self.supervise_single:
            c = args['in_head_single']
            self.cls_head_single = nn.Conv2d(c, args['anchor_number'], kernel_size=1)
            self.reg_head_single = nn.Conv2d(c, args['anchor_number'] * 7, kernel_size=1)
            self.dir_head_single = nn.Conv2d(c, args['anchor_number'] * args['dir_args']['num_bins'], kernel_size=1)

        if self.fusion_method not in _FUSIONS:
            raise NotImplementedError(f"fusion_method {self.fusion_method!r} is outside the hot path (att / max)")
        self.fusion_net = _FUSIONS[self.fusion_method](args)

        self.shrink_flag = 'shrink_header' in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args['shrink_header'])

        n_cls, n_reg, n_dir = self._head_widths(args)
        self.cls_head = nn.Conv2d(args['in_head'], n_cls, kernel_size=1)
        self.reg_head = nn.Conv2d(args['in_head'], n_reg, kernel_size=1)
        self.dir_head = nn.Conv2d(args['in_head'], n_dir, kernel_size=1)

        self.compress = 'compressor' in args
        if self.compress:
            self.compressor = NaiveCompressor(args['compressor']['input_dim'], args['compressor']['compress_ratio'])
            self.model_train_init()

    @staticmethod
    def _head_widths(args):
        a = args['anchor_number']
        return a, 7 * a, args['dir_args']['num_bins'] * a

    def model_train_init(self):
        if self.compress:  # only the compressor stays trainable
            self.eval()
            for p in self.parameters():
                p.requires_grad_(False)
            self.compressor.train()
            for p in self.compressor.parameters():
                p.requires_grad_(True)

    def get_memory_footprint(self):
        total = sum(t.nelement() * t.element_size() for t in list(self.parameters()) + list(self.buffers()))
        return f"Model Memory Footprint: {total / (1024 ** 2):.2f} MB"

    # ---- forward, in stages so that subclasses / the deployed engine can reuse them ----
    def encode_agents(self, data_dict):
        """Per-agent features in ``agent_modality_list`` order: [sum_N, C, H, W]."""
        agents = data_dict['agent_modality_list']
        present = Counter(agents)
        per_mod = {}
        for name in self.modality_name_list:
            if name in present:
                f = getattr(self, f"encoder_{name}")(data_dict, name)
                f = getattr(self, f"backbone_{name}")(f)
                per_mod[name] = getattr(self, f"shrinker_{name}")(f)
        taken = {name: 0 for name in self.modality_name_list}
        rows = []
        for name in agents:
            rows.append(per_mod[name][taken[name]])
            taken[name] += 1
        return torch.stack(rows)

    def transform_features(self, feats, output_dict):
        """Hook between encode and fusion (compressor here, codebook in the subclass)."""
        return self.compressor(feats) if self.compress else feats

    def forward(self, data_dict):
        out = {}
        affine = normalize_pairwise_tfm(data_dict['pairwise_t_matrix'], self.H, self.W, self.fake_voxel_size)
        feats = self.transform_features(self.encode_agents(data_dict), out)
        if self.supervise_single:
            out.update({'cls_preds_single': self.cls_head_single(feats),
                        'reg_preds_single': self.reg_head_single(feats),
                        'dir_preds_single': self.dir_head_single(feats)})
        fused = self.fusion_net(feats, data_dict['record_len'], affine)
        if self.shrink_flag:
            fused = self.shrink_conv(fused)
        cls, reg, dr = self.cls_head(fused), self.reg_head(fused), self.dir_head(fused)
        out.update({'cls_preds': cls, 'reg_preds': reg, 'dir_preds': dr,
                    'preds_tensor': torch.cat([cls, reg, dr], dim=1)})
        return out
