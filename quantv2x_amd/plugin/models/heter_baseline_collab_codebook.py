"""Baseline + residual multi-codebook compressor on the shared BEV feature; mirror of
``opencood/models/heter_baseline_collab_codebook.py`` (ctor ``:44-57``, codebook step ``:119-132``).

``args['codebook'] = {seg_num, dict_size}``; three residual levels, C = 256.
Extra key (this build): ``args['codebook']['hard_eval']`` -- when true and the module is in
eval mode the feature goes through the deterministic ``encode -> decode`` pair (the wire
format the deployed engine all-gathers) instead of the reference's Gumbel ``forward``,
which draws random numbers even in eval (SURVEY.md §7 "Soft path is stochastic in eval").
"""
import torch
import torch.nn as nn

from .heter_model_baseline import HeterModelBaseline
from .sub_modules.codebook import UMGMQuantizer


class _CodebookMixin:
    def _build_codebook(self, args):
        self.channel = 256
        cb = args.get('codebook')
        if cb is not None:
            self.seg_num = cb['seg_num']
            self.dict_size = [cb['dict_size']] * 3
        else:
            self.seg_num, self.dict_size = 2, [256] * 3
        self.hard_eval = bool(cb.get('hard_eval', False)) if cb is not None else False
        self.p_rate = 0.0
        c = self.channel
        heads = ("latentStageEncoder", "quantizationHead", "latentHead",
                 "restoreHead", "dequantizationHead", "sideHead")
        self.codebook = UMGMQuantizer(c, self.seg_num, self.dict_size, self.p_rate,
                                      {h: (lambda: nn.Linear(c, c)) for h in heads})

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


class HeterBaselineCollabCodebook(_CodebookMixin, HeterModelBaseline):
    def __init__(self, args):
        super().__init__(args)
        self._build_codebook(args)
