"""Activation-quantizer initialisation; mirror of ``opencood/quant/set_act_quantize_params.py:7-32``.

The reference moves each cached input with ``.cuda()``; here the input follows the module's device, so the
routine also runs on the CPU container; sparse wrappers are included as in ``second_recon.py:33-50``; the
routine also runs without a GPU (on a ROCm box ``cuda`` is the HIP device either way)."""
from typing import Union

import torch

from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule, QuantSpconvModule
from .quant_model import QuantModel


def _module_device(module):
    for p in module.parameters():
        return p.device
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _move(x, dev):
    if isinstance(x, dict):
        return {k: _move(v, dev) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_move(v, dev) for v in x)
    return x.to(dev) if hasattr(x, "to") else x


def set_act_quantize_params(module: Union[QuantModel, QuantModule, BaseQuantBlock],
                            cached_inps: Union[list, torch.Tensor], channel_sizes: list = None, batch_size: int = 8, extras: list = None):
    """``extras``: per cached input, the block's further positional inputs (``pyramid_recon.set_act_quantize_params :104-121``)."""
    module.set_quant_state(True, True)
    holders = [t for t in module.modules()
               if isinstance(t, (QuantSpconvModule, QuantModule, BaseQuantBlock)) and hasattr(t, 'act_quantizer')]
    for t in holders:
        t.act_quantizer.set_inited(False)
    dev = _module_device(module)
    n = cached_inps.size(0) if isinstance(cached_inps, torch.Tensor) else min(len(cached_inps), batch_size)
    with torch.no_grad():
        for i in range(n):
            module(_move(cached_inps[i], dev), *(_move(tuple(extras[i]), dev) if extras is not None else ()))
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    for t in holders:
        t.act_quantizer.set_inited(True)
