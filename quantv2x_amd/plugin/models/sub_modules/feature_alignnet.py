"""``AlignNet`` (``opencood/models/sub_modules/feature_alignnet.py:12-39``): the LiDAR-only yamls use ``core_method: identity``;
the learned aligners belong to the heterogeneous-modality training recipes and are outside the hot path."""
import torch.nn as nn


class AlignNet(nn.Module):
    def __init__(self, args):
        super().__init__()
        if args['core_method'] != 'identity':
            raise NotImplementedError(f"aligner {args['core_method']!r} is outside the hot path (identity only)")
        self.channel_align = nn.Identity()
        if args.get("spatial_align", False):
            raise NotImplementedError

    def forward(self, x):
        return self.channel_align(x)
