"""PTQ observer / fake-quant surface; host-side mirror of ``opencood/quant/quant_layer.py``.

* ``UniformAffineQuantizer`` (reference ``:53-346``): asymmetric *unsigned* n-bit affine quantizer,
  ``x_q = clamp(round(x / delta) + zp, 0, 2^n - 1)``, ``x' = (x_q - zp) * delta``; per-dim-0 scales
  when ``channel_wise`` (weights -- for ``ConvTranspose2d`` dim 0 is C_in, a quirk that is kept),
  per-tensor otherwise; range init by min/max, MSE (p = 2.4) 1-D / 2-D search or entropy; EMA
  0.9/0.1 running range when ``leaf_param``.
* ``QuantModule`` (reference ``:349-420``): conv / deconv / linear evaluated on fake-quantized
  weights, then attached norm + activation, then the output activation quantizer.

These run in torch (autograd-capable) and exist for calibration.  The frozen W8A8 state they hold
(delta, zero_point, integer weight codes) is what ``quantv2x_amd.deploy`` compiles into the HIP int8 path.
"""
from typing import Union

import torch
import torch.nn as nn
import torch.nn.functional as F


class StraightThrough(nn.Module):
    def forward(self, input):
        return input


def round_ste(x: torch.Tensor):
    """round() with identity gradient."""
    return (x.round() - x).detach() + x


def lp_loss(pred, tgt, p=2.0, reduction='none'):
    err = (pred - tgt).abs().pow(p)
    return err.sum(1).mean() if reduction == 'none' else err.mean()


class UniformAffineQuantizer(nn.Module):
    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False,
                 scale_method: str = 'mse', leaf_param: bool = False, prob: float = 1.0):
        super().__init__()
        self.sym = symmetric
        if self.sym:
            raise NotImplementedError
        assert 2 <= n_bits <= 8, 'bitwidth not supported'
        self.n_bits = n_bits
        self.n_levels = 2 ** n_bits
        self.delta = 1.0
        self.zero_point = 0.0
        self.inited = True
        self.leaf_param = leaf_param        # activation quantizer: EMA over observed ranges
        self.channel_wise = channel_wise
        self.eps = torch.tensor(1e-8, dtype=torch.float32)
        self.scale_method = scale_method
        self.one_side_dist = None           # 'pos' | 'neg' | 'no', decided on first observation
        self.num = 100                      # grid points of the MSE search
        self.running_min = None
        self.running_max = None
        self.prob = prob                    # QDrop keep probability
        self.is_training = False

    def set_inited(self, inited: bool = True):
        self.inited = inited

    def bitwidth_refactor(self, refactored_bit: int):
        assert 2 <= refactored_bit <= 8, 'bitwidth not supported'
        self.n_bits = refactored_bit
        self.n_levels = 2 ** refactored_bit

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if self.inited is False:
            self.delta, self.zero_point = self.init_quantization_scale(x.clone().detach(), self.channel_wise)
        else:
            # (delta, zero_point) are plain attributes, not buffers: ``model.cuda()`` after the weight ranges were taken on the CPU
            # leaves them behind (the reference builds on the GPU from the start and never meets this)
            for name in ("delta", "zero_point"):
                t = getattr(self, name)
                if isinstance(t, torch.Tensor) and not isinstance(t, torch.nn.Parameter) and t.device != x.device:
                    setattr(self, name, t.to(x.device))
        code = torch.clamp(round_ste(x / self.delta) + self.zero_point, 0, self.n_levels - 1)
        deq = (code - self.zero_point) * self.delta
        if self.is_training and self.prob < 1.0:
            return torch.where(torch.rand_like(x) < self.prob, deq, x)
        return deq

    # ---- range -> (delta, zero_point) --------------------------------------------------------
    def calculate_qparams(self, min_val, max_val):
        lo = torch.min(min_val, torch.zeros_like(min_val))
        hi = torch.max(max_val, torch.zeros_like(max_val))
        scale = torch.max((hi - lo) / float(self.n_levels - 1), self.eps)
        zero_point = torch.clamp(0 - torch.round(lo / scale), 0, self.n_levels - 1)
        return scale, zero_point

    def _per_channel_view(self, t, like):
        shape = [1] * like.dim()
        shape[0] = like.shape[0]
        return t.reshape(shape)

    def quantize(self, x: torch.Tensor, x_max, x_min):
        delta, zp = self.calculate_qparams(x_min, x_max)
        if self.channel_wise:
            delta, zp = self._per_channel_view(delta, x), self._per_channel_view(zp, x)
        code = torch.clamp(torch.round(x / delta) + zp, 0, self.n_levels - 1)
        return (code - zp) * delta

    def lp_loss(self, pred, tgt, p=2.0):
        err = (pred - tgt).abs().pow(p)
        return torch.flatten(err, 1).mean(1) if self.channel_wise else err.mean()

    def update_quantize_range(self, x_min, x_max):
        if self.running_min is None:
            self.running_min, self.running_max = x_min, x_max
        self.running_min = 0.1 * x_min + 0.9 * self.running_min
        self.running_max = 0.1 * x_max + 0.9 * self.running_max
        return self.running_min, self.running_max

    def _observed_range(self, x, include_zero):
        if self.channel_wise:
            lo, hi = torch.aminmax(torch.flatten(x, 1), dim=1)
            if include_zero:
                hi = torch.max(hi, torch.zeros_like(hi))
                lo = torch.min(lo, torch.zeros_like(lo))
            return lo, hi
        return tuple(torch.aminmax(x))

    def perform_2D_search(self, x):
        """Search (range width, zero point) minimising the L2.4 error (reference ``:202-234``)."""
        x_min, x_max = self._observed_range(x, include_zero=True)
        if self.scale_method == 'minmax':
            return x_min, x_max
        span = x_max - x_min
        best = torch.zeros_like(x_min) + 1e+10
        best_min, best_max = x_min.clone(), x_max.clone()
        for i in range(1, self.num + 1):
            top = span / self.num * i
            step = (top - torch.zeros_like(x_min)) / (2 ** self.n_bits - 1)
            for zp in range(0, self.n_levels):
                lo, hi = torch.zeros_like(x_min) - zp * step, top - zp * step
                score = self.lp_loss(x, self.quantize(x, hi, lo), 2.4)
                best_min = torch.where(score < best, lo, best_min)
                best_max = torch.where(score < best, hi, best_max)
                best = torch.min(best, score)
        return best_min, best_max

    def perform_1D_search(self, x):
        """One-sided or symmetric threshold search (reference ``:236-258``)."""
        x_min, x_max = self._observed_range(x, include_zero=False)
        if self.scale_method == 'minmax':
            return x_min, x_max
        reach = torch.max(x_min.abs(), x_max)
        best = torch.zeros_like(x_min) + 1e+10
        best_min, best_max = x_min.clone(), x_max.clone()
        for i in range(1, self.num + 1):
            thr = reach / self.num * i
            lo = torch.zeros_like(x_min) if self.one_side_dist == 'pos' else -thr
            hi = torch.zeros_like(x_max) if self.one_side_dist == 'neg' else thr
            score = self.lp_loss(x, self.quantize(x, hi, lo), 2.4)
            best_min = torch.where(score < best, lo, best_min)
            best_max = torch.where(score < best, hi, best_max)
            best = torch.min(score, best)
        return best_min, best_max

    def perform_entropy_search(self, x, num_bins=2048, num_quant_bins=None):
        """KL-minimising clip of the upper range over a 2048-bin histogram (reference ``:276-321``)."""
        if num_quant_bins is None:
            num_quant_bins = self.n_levels
        x = x.detach().float()
        if self.channel_wise:
            raise NotImplementedError("Channel-wise entropy search is not yet supported.")
        x_min, x_max = x.min(), x.max()
        hist = torch.histc(x, bins=num_bins, min=x_min.item(), max=x_max.item())
        width = (x_max - x_min) / num_bins
        best_kl, best_max = float('inf'), x_max
        for i in range(num_bins // 2, num_bins):
            ratio = i // num_quant_bins
            if ratio < 1:
                continue
            clipped = hist.clone()
            clipped[i:] = clipped[i - 1:].sum()
            coarse = clipped[:i].reshape(num_quant_bins, ratio).sum(dim=1).repeat_interleave(ratio)
            if coarse.shape[0] < i:
                coarse = F.pad(coarse, (0, i - coarse.shape[0]))
            p = clipped[:i]
            q = coarse + 1e-6
            p, q = p / p.sum(), q / q.sum()
            kl = (p * (p / q).log()).sum()
            if kl < best_kl:
                best_kl, best_max = kl, x_min + width * i
        return x_min, best_max

    def get_x_min_x_max(self, x):
        if self.scale_method not in ['mse', 'minmax', 'entropy']:
            raise NotImplementedError
        if self.scale_method == 'entropy':
            lo, hi = self.perform_entropy_search(x)
        elif self.one_side_dist is None:
            self.one_side_dist = 'pos' if x.min() >= 0.0 else 'neg' if x.max() <= 0.0 else 'no'
            lo, hi = self.perform_1D_search(x) if (self.one_side_dist != 'no' or self.sym) \
                else self.perform_2D_search(x)
        else:
            lo, hi = self.perform_2D_search(x)
        if self.leaf_param:
            return self.update_quantize_range(lo, hi)
        return lo, hi

    def init_quantization_scale_channel(self, x: torch.Tensor):
        return self.calculate_qparams(*self.get_x_min_x_max(x))

    def init_quantization_scale(self, x_clone: torch.Tensor, channel_wise: bool = False):
        delta, zp = self.init_quantization_scale_channel(x_clone)
        if channel_wise:
            delta, zp = self._per_channel_view(delta, x_clone), self._per_channel_view(zp, x_clone)
        return delta, zp

    @torch.jit.export
    def extra_repr(self):
        return 'bit={}, is_training={}, inited={}'.format(self.n_bits, self.is_training, self.inited)


# torch functional + the constructor attributes it takes over from the wrapped module, by module type (first match wins)
_FWD_TABLE = (
    (nn.Conv2d, F.conv2d, ("stride", "padding", "dilation", "groups")),
    (nn.ConvTranspose2d, F.conv_transpose2d, ("stride", "padding", "output_padding", "groups", "dilation")),
    (nn.Linear, F.linear, ()),
)


class QuantModule(nn.Module):
    """One conv / deconv / linear under fake quantization -- the reference's ``QuantModule`` (quant_layer.py:349-420) as far as its
    users see it: the attribute set below is a contract (``quant_model.py``, ``fold_bn.py``, the ``*_recon.py`` loops and
    ``inference_quant.py`` read and assign ``weight, org_weight, bias, org_bias, fwd_func, fwd_kwargs, use_weight_quant, use_act_quant,
    weight_quantizer, act_quantizer, norm_function, activation_function, ignore_reconstruction, disable_act_quant, trained``).

    forward = ``act_quantizer(activation(norm(op(x, W', b'))))`` with ``(W', b')`` the fake-quantized pair while weight quantization is
    on and the untouched originals otherwise; the output quantizer is skipped while activation quantization is off or the layer feeds an
    element-wise sum (``disable_act_quant``: the block quantizes after the sum)."""

    def __init__(self, org_module: Union[nn.Conv2d, nn.ConvTranspose2d, nn.Linear], weight_quant_params: dict = {},
                 act_quant_params: dict = {}, disable_act_quant=False):
        super().__init__()
        for kind, func, attrs in _FWD_TABLE:
            if isinstance(org_module, kind):
                self.fwd_func, self.fwd_kwargs = func, {a: getattr(org_module, a) for a in attrs}
                break
        else:                                                   # (the reference sends everything else down the linear path too)
            self.fwd_func, self.fwd_kwargs = F.linear, {}
        has_bias = org_module.bias is not None
        self.weight, self.org_weight = org_module.weight, org_module.weight.data.clone()
        self.bias, self.org_bias = (org_module.bias, org_module.bias.data.clone()) if has_bias else (None, None)
        self.weight_quantizer = UniformAffineQuantizer(**weight_quant_params)
        self.act_quantizer = UniformAffineQuantizer(**act_quant_params)
        self.norm_function, self.activation_function = StraightThrough(), StraightThrough()   # fold_bn / the wrapper blocks fill these in
        self.set_quant_state(False, False)                      # a fresh wrapper runs the original arithmetic
        self.ignore_reconstruction, self.trained = False, False
        self.disable_act_quant = disable_act_quant

    def _effective_params(self, device):
        """(weight, bias) the op runs on right now, on ``device``"""
        if self.use_weight_quant:
            w, b = self.weight_quantizer(self.weight), self.bias
        else:
            w, b = self.org_weight, self.org_bias
        return w.to(device), (b.to(device) if b is not None else None)

    def _normalize(self, out):
        if type(self.norm_function) is nn.BatchNorm1d:          # pillar layout [M, P, C]: BatchNorm1d wants the channels second
            return self.norm_function(out.transpose(1, 2)).transpose(1, 2)
        return self.norm_function(out)

    def forward(self, input: torch.Tensor):
        out = self.activation_function(self._normalize(self.fwd_func(input, *self._effective_params(input.device), **self.fwd_kwargs)))
        quantize_output = self.use_act_quant and not self.disable_act_quant
        return self.act_quantizer(out) if quantize_output else out

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant, self.use_act_quant = weight_quant, act_quant

    @torch.jit.export
    def extra_repr(self):
        return f"wbit={self.weight_quantizer.n_bits}, abit={self.act_quantizer.n_bits}, disable_act_quant={self.disable_act_quant}"


class QuantSpconvModule(nn.Module):
    """A sparse convolution evaluated on fake-quantized weights, then its BatchNorm1d / ReLU on the feature rows, then the output
    activation quantizer (reference ``quant_layer.py:423-497``).  BatchNorm is NOT folded here: ``fold_bn`` only absorbs into
    ``Conv2d`` / ``Linear``, so the deployed path keeps the per-channel affine after the integer accumulation."""

    def __init__(self, org_module, weight_quant_params={}, act_quant_params={}, disable_act_quant=False):
        super().__init__()
        from ..models.sub_modules.sparse_ops import SparseConv3d, SubMConv3d
        assert isinstance(org_module, (SubMConv3d, SparseConv3d)), "QuantSpconvModule wraps SubMConv3d / SparseConv3d"
        self.spconv_module = org_module
        self.weight = org_module.weight
        self.org_weight = org_module.weight.data.clone()
        self.bias = org_module.bias
        self.org_bias = None if org_module.bias is None else org_module.bias.data.clone()
        self.use_weight_quant = False
        self.use_act_quant = False
        self.disable_act_quant = disable_act_quant
        self.ignore_reconstruction = False
        self.trained = False
        self.weight_quantizer = UniformAffineQuantizer(**weight_quant_params)
        self.act_quantizer = UniformAffineQuantizer(**act_quant_params)
        self.norm_function = StraightThrough()
        self.activation_function = StraightThrough()
        self.is_sparse_conv = True

    def forward(self, input):
        weight = self.weight_quantizer(self.weight) if self.use_weight_quant else self.org_weight
        bias = self.org_bias
        dev = input.features.device
        out = self.spconv_module(input, weight=weight.to(dev), bias=None if bias is None else bias.to(dev))
        out = out.replace_feature(self.activation_function(self.norm_function(out.features)))
        if self.use_act_quant and not self.disable_act_quant:
            out = out.replace_feature(self.act_quantizer(out.features))
        return out

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant

    def extra_repr(self):
        return f"wbit={self.weight_quantizer.n_bits}, abit={self.act_quantizer.n_bits}, disable_act_quant={self.disable_act_quant}"
