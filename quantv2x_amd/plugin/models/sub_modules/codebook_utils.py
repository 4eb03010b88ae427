"""Helpers of the residual multi-codebook quantizer; mirror of
``opencood/models/sub_modules/codebook_utils.py`` (LowerBound ``:19-57``, gumbelSoftmax ``:60-76``, CodeSize ``:79-99``)."""
from dataclasses import dataclass
from typing import List

import torch
from torch import nn


class _ClampBelow(torch.autograd.Function):
    """max(x, bound) whose gradient passes when x is above the bound or is being pushed up."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (g < 0)
        return keep.type(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _ClampBelow.apply(x, self.bound)


def gumbelSoftmax(logits: torch.Tensor, temperature: float = 1.0, hard: bool = True, dim: int = -1):
    """Gumbel-perturbed softmax; with ``hard`` a straight-through one-hot.  Draws
    ``torch.rand_like(logits)`` exactly once, so under one ``torch.manual_seed`` it
    consumes the generator like the reference does."""
    tiny = torch.finfo(logits.dtype).eps
    u = torch.rand_like(logits).clamp_(tiny, 1 - tiny)
    g = -((-(u.log())).log())
    soft = ((logits + g) / temperature).softmax(dim)
    if not hard:
        return soft
    top = soft.max(dim, keepdim=True)[1]
    one_hot = torch.zeros_like(logits, memory_format=torch.legacy_contiguous_format).scatter_(dim, top, 1.0)
    return one_hot - soft.detach() + soft


@dataclass
class CodeSize:
    m: int
    heights: List[int]
    widths: List[int]
    k: List[int]

    def __str__(self) -> str:
        seq = ", ".join(f"[{w}x{h}, {k}]" for h, w, k in zip(self.heights, self.widths, self.k))
        return f"\n        {self.m} code-groups: {seq}"
