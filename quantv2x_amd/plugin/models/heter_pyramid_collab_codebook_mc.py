"""Pyramid model + residual multi-codebook compressor on the 64-channel per-agent BEV feature; mirror of
``opencood/models/heter_pyramid_collab_codebook_mc.py`` (ctor ``:25-51``, codebook step ``:113-127``): ``channel = 64``,
``args['codebook'] = {seg_num, dict_size}`` (default 2 / 256), three residual levels, no compressor on this path.

``args['codebook']['hard_eval']`` (this build, as in ``heter_baseline_collab_codebook``): in eval mode use the deterministic
``encode -> decode`` pair instead of the Gumbel ``forward`` (which draws random numbers even in eval)."""
import torch
import torch.nn as nn

from .heter_pyramid_collab_mc import HeterPyramidCollabMC
from .sub_modules.codebook import UMGMQuantizer


class HeterPyramidCollabCodebookMC(HeterPyramidCollabMC):
    def __init__(self, args):
        super().__init__(args)
        self.channel = 64
        cb = args.get('codebook')
        if cb is not None:
            self.seg_num, self.dict_size = cb['seg_num'], [cb['dict_size']] * 3
        else:
            self.seg_num, self.dict_size = 2, [256] * 3
        self.hard_eval = bool(cb.get('hard_eval', False)) if cb is not None else False
        self.p_rate = 0.0
        c = self.channel
        heads = ("latentStageEncoder", "quantizationHead", "latentHead", "restoreHead", "dequantizationHead", "sideHead")
        self.codebook = UMGMQuantizer(c, self.seg_num, self.dict_size, self.p_rate, {h: (lambda: nn.Linear(c, c)) for h in heads})

    def transform_features(self, feats, output_dict):
        n, c, h, w = feats.shape
        rows = feats.permute(0, 2, 3, 1).contiguous().view(-1, c)
        if self.hard_eval and not self.training:
            restored = self.codebook.decode(self.codebook.encode(rows))
            loss = torch.nn.functional.mse_loss(restored, rows)
        else:
            restored, _, _, loss = self.codebook(rows)
        output_dict['codebook_loss'] = loss
        return restored.view(n, h, w, c).permute(0, 3, 1, 2).contiguous()
