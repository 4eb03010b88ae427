"""Residual multi-codebook feature compressor (UMGM quantizer, after McQuic).

Host-side mirror of ``opencood/models/sub_modules/codebook.py``: same class and
attribute names, hence the same ``state_dict`` keys
(``_encoders.{l}._quantizer._codebook [m,k,d]``, ``._temperature``, ``._bound.bound``,
``._latentStageEncoder``, ``._quantizationHead``, ``._latentHead``;
``_decoders.{l}._dequantizationHead``, ``._sideHead``, ``._restoreHead``; ``_freqEMA.{l}``).

Per level l (reference ``_quantizerEncoder.encode :231-239``, ``_distance :115-131``):
    z = latentStageEncoder(x);  q = quantizationHead(z)
    code = argmin_k ( |q|^2 + |C_k|^2 - 2 q.C_k )
    x <- latentHead(z) - C[code]                 (no latentHead on the last level)
Decode, levels reversed (``_quantizerDecoder.decode :263-269``):
    out = restoreHead( dequantizationHead(C[code]) + sideHead(previous) )

``encode``/``decode`` are deterministic and are what the deployed HIP path
(``qv2x_codebook_encode_f32`` / the decode LUT in ``qv2x_fuse_*``) implements.
``forward`` is the reference's soft Gumbel path (``:375-408``), stochastic even in eval.
"""
import math
from typing import Callable, Dict, List, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from .codebook_utils import CodeSize, LowerBound, gumbelSoftmax

Eps = 1e-6


class BaseQuantizer(nn.Module):
    def __init__(self, m: int, k: List[int]):
        super().__init__()
        self._m = m
        self._k = k

    def encode(self, x):
        raise NotImplementedError

    def decode(self, codes):
        raise NotImplementedError

    @property
    def Codebooks(self):
        raise NotImplementedError


class _multiCodebookQuantization(nn.Module):
    def __init__(self, codebook: nn.Parameter, permutationRate: float = 0.0):
        super().__init__()
        self._m, self._k, self._d = codebook.shape
        self._codebook = codebook
        self._scale = math.sqrt(self._k)
        self._temperature = nn.Parameter(torch.ones((self._m, 1)))
        self._bound = LowerBound(Eps)
        self._permutationRate = permutationRate

    def _distance(self, x: torch.Tensor) -> torch.Tensor:
        """[n, m*d] -> squared distances [n, m, k] as |x|^2 + |c|^2 - 2 x.c (this op order)."""
        xs = x.reshape(x.shape[0], self._m, self._d)
        x2 = (xs ** 2).sum(2, keepdim=True)
        c2 = (self._codebook ** 2).sum(-1, keepdim=False)
        cross = torch.einsum("nmd,mkd->nmk", xs, self._codebook)
        return x2 + c2 - 2 * cross

    def encode(self, x: torch.Tensor):
        return self._distance(x).argmin(-1)

    def _logit(self, x):
        return -1 * self._distance(x) / self._scale

    def _sample(self, x, temperature: float):
        logit = self._logit(x) * self._bound(self._temperature)
        if self._permutationRate >= Eps:
            raise NotImplementedError("code permutation is a training-time augmentation (outside the inference path)")
        return gumbelSoftmax(logit, temperature, True), logit

    def forward(self, x):
        sample, logit = self._sample(x, 1.0)
        code = logit.argmax(-1, keepdim=True)
        one_hot = torch.zeros_like(logit).scatter_(-1, code, 1)
        return sample, code[..., 0], one_hot, logit


class _multiCodebookDeQuantization(nn.Module):
    def __init__(self, codebook: nn.Parameter):
        super().__init__()
        self._m, self._k, self._d = codebook.shape
        self._codebook = codebook
        self.register_buffer("_ix", torch.arange(self._m), persistent=False)

    def decode(self, code: torch.Tensor):
        picked = self._codebook[self._ix.expand_as(code), code]  # [n, m, d]
        return picked.reshape(code.shape[0], -1)

    def forward(self, sample: torch.Tensor):
        return torch.einsum("nmk,mkd->nmd", sample, self._codebook).reshape(sample.shape[0], -1)


class _quantizerEncoder(nn.Module):
    def __init__(self, quantizer, dequantizer, latentStageEncoder, quantizationHead, latentHead):
        super().__init__()
        self._quantizer = quantizer
        self._dequantizer = dequantizer
        self._latentStageEncoder = latentStageEncoder
        self._quantizationHead = quantizationHead
        self._latentHead = latentHead

    @property
    def Codebook(self):
        return self._quantizer._codebook

    def encode(self, x):
        z = self._latentStageEncoder(x)
        code = self._quantizer.encode(self._quantizationHead(z))
        if self._latentHead is None:
            return None, code
        return self._latentHead(z) - self._dequantizer.decode(code), code

    def forward(self, x):
        z = self._latentStageEncoder(x)
        q, code, one_hot, logit = self._quantizer(self._quantizationHead(z))
        if self._latentHead is None:
            return q, None, code, one_hot, logit
        return q, self._latentHead(z) - self._dequantizer(q), code, one_hot, logit


class _quantizerDecoder(nn.Module):
    def __init__(self, dequantizer, dequantizationHead, sideHead, restoreHead):
        super().__init__()
        self._dequantizer = dequantizer
        self._dequantizationHead = dequantizationHead
        self._sideHead = sideHead
        self._restoreHead = restoreHead

    def _merge(self, q, former):
        return self._restoreHead(q if self._sideHead is None else q + self._sideHead(former))

    def decode(self, code, formerLevel):
        return self._merge(self._dequantizationHead(self._dequantizer.decode(code)), formerLevel)

    def forward(self, q, formerLevel):
        return self._merge(self._dequantizationHead(self._dequantizer(q)), formerLevel)


class UMGMQuantizer(BaseQuantizer):
    _components = ["latentStageEncoder", "quantizationHead", "latentHead",
                   "dequantizationHead", "sideHead", "restoreHead"]

    def __init__(self, channel: int, m: int, k: Union[int, List[int]], permutationRate: float,
                 components: Dict[str, Callable[[], nn.Module]]):
        if isinstance(k, int):
            k = [k]
        super().__init__(m, k)
        self.ema = 0.9
        self._freqEMA = nn.ParameterList(nn.Parameter(torch.ones(m, ki) / ki, requires_grad=False) for ki in k)
        mk = {name: components[name] for name in self._components}
        encoders, decoders = [], []
        for lvl, ki in enumerate(k):
            last = lvl == len(k) - 1
            # construction order of the heads follows the reference (:297-302) so torch's RNG is consumed alike
            stage = mk["latentStageEncoder"]()
            qhead = mk["quantizationHead"]()
            lhead = None if last else mk["latentHead"]()
            dqhead = mk["dequantizationHead"]()
            side = None if last else mk["sideHead"]()
            restore = mk["restoreHead"]()
            book = nn.Parameter(nn.init.normal_(torch.empty(m, ki, channel // m),
                                                std=math.sqrt(2 / (5 * channel / m))))
            quantizer = _multiCodebookQuantization(book, permutationRate)
            dequantizer = _multiCodebookDeQuantization(book)
            encoders.append(_quantizerEncoder(quantizer, dequantizer, stage, qhead, lhead))
            decoders.append(_quantizerDecoder(dequantizer, dqhead, side, restore))
        self._encoders = nn.ModuleList(encoders)
        self._decoders = nn.ModuleList(decoders)

    @property
    def Codebooks(self):
        return [e.Codebook for e in self._encoders]

    def encode(self, x: torch.Tensor) -> List[torch.Tensor]:
        codes = []
        for enc in self._encoders:
            x, code = enc.encode(x)
            codes.append(code)
        return codes

    def decode(self, codes: List[torch.Tensor]):
        out = None
        for dec, code in zip(self._decoders[::-1], codes[::-1]):
            out = dec.decode(code, out)
        return out

    def updateFreq(self, onehot_list):
        for lvl, one_hot in enumerate(onehot_list):
            hist = one_hot.sum(0)
            self._freqEMA[lvl] = (1 - self.ema) * (hist / hist.sum(-1, keepdim=True)) + self.ema * self._freqEMA[lvl]

    def normalFreq(self):
        return [(f / f.sum(-1, keepdim=True)).clone().detach() for f in self._freqEMA]

    def forward(self, x: torch.Tensor):
        target = x.detach()
        soft, codes, one_hots, logits = [], [], [], []
        for enc in self._encoders:
            q, x, code, one_hot, logit = enc(x)
            soft.append(q); codes.append(code); one_hots.append(one_hot); logits.append(logit)
        out = None
        for dec, q in zip(self._decoders[::-1], soft[::-1]):
            out = dec(q, out)
        self.updateFreq(one_hots)
        return out, codes, logits, F.mse_loss(out, target)
