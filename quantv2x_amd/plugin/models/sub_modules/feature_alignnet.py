"""``AlignNet`` (``opencood/models/sub_modules/feature_alignnet.py:12-39``): the LiDAR-only yamls use ``core_method: identity``;
the learned aligners belong to the heterogeneous-modality training recipes and are outside the hot path."""
from torch import nn


class AlignNet(nn.Module):
    """Per-modality feature aligner between the agent backbone and the fusion; attribute ``channel_align`` as in the reference."""

    def __init__(self, aligner_cfg: dict):
        nn.Module.__init__(self)
        method = aligner_cfg['core_method']
        if method != 'identity' or aligner_cfg.get("spatial_align", False):
            raise NotImplementedError(f"aligner {method!r} (spatial_align={aligner_cfg.get('spatial_align', False)}) is outside the hot path: identity only")
        self.channel_align = nn.Identity()

    def forward(self, feature):
        return self.channel_align(feature)
