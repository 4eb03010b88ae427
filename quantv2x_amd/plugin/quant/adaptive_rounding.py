"""AdaRound weight quantizer; mirror of ``opencood/quant/adaptive_rounding.py:6-74``.

``learned_hard_sigmoid``: ``w_q = floor(w / delta) + h(alpha)`` with the rectified sigmoid
``h = clip(sigmoid(alpha) * (zeta - gamma) + gamma, 0, 1)`` while training (``soft_targets``) and the
hard mask ``alpha >= 0`` afterwards -- the mask the deployed int8 weights are built from."""
import torch
from torch import nn

from .quant_layer import UniformAffineQuantizer, round_ste


class AdaRoundQuantizer(nn.Module):
    def __init__(self, uaq: UniformAffineQuantizer, weight_tensor: torch.Tensor, round_mode='learned_round_sigmoid'):
        super().__init__()
        self.n_bits, self.sym = uaq.n_bits, uaq.sym
        self.delta, self.zero_point, self.n_levels = uaq.delta, uaq.zero_point, uaq.n_levels
        self.round_mode = round_mode
        self.alpha = None
        self.soft_targets = False
        self.gamma, self.zeta = -0.1, 1.1
        self.beta = 2 / 3
        self.init_alpha(x=weight_tensor.clone())

    def forward(self, x):
        scaled = x / self.delta
        if self.round_mode == 'nearest':
            code = torch.round(scaled)
        elif self.round_mode == 'nearest_ste':
            code = round_ste(scaled)
        elif self.round_mode == 'stochastic':
            base = torch.floor(scaled)
            code = base + torch.bernoulli(scaled - base)
        elif self.round_mode == 'learned_hard_sigmoid':
            up = self.get_soft_targets() if self.soft_targets else (self.alpha >= 0).float()
            code = torch.floor(scaled) + up
        else:
            raise ValueError('Wrong rounding mode')
        code = torch.clamp(code + self.zero_point, 0, self.n_levels - 1)
        return (code - self.zero_point) * self.delta

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def init_alpha(self, x: torch.Tensor):
        if self.round_mode != 'learned_hard_sigmoid':
            raise NotImplementedError
        frac = x / self.delta - torch.floor(x / self.delta)
        # sigmoid(alpha) stretched to (gamma, zeta) equals the fractional part
        self.alpha = nn.Parameter(-torch.log((self.zeta - self.gamma) / (frac - self.gamma) - 1))

    @torch.jit.export
    def extra_repr(self):
        return 'bit={}'.format(self.n_bits)
