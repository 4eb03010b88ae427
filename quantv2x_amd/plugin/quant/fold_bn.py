"""BatchNorm folding; mirror of ``opencood/quant/fold_bn.py`` (``_fold_bn :19-127``,
``search_fold_and_remove_bn :161-175``).

    W' = W * gamma / sigma        (broadcast over C_out: dim 0 for Conv2d / Linear, dim 1 for ConvTranspose2d)
    b' = beta - gamma * mu / sigma  (+ gamma * b / sigma when the layer had a bias),  sigma = sqrt(var + eps)
"""
import torch
import torch.nn as nn

from .quant_layer import StraightThrough  # noqa: F401  (same class the reference's module exposes)

_UNTOUCHED = None  # filled lazily from quant_block to avoid an import cycle


def _skip_names():
    global _UNTOUCHED
    if _UNTOUCHED is None:
        from .quant_block import specials_unquantized_names
        _UNTOUCHED = specials_unquantized_names
    return _UNTOUCHED


def _fold_bn(conv_module, bn_module):
    w = conv_module.weight.data
    sigma = torch.sqrt(bn_module.running_var + bn_module.eps)
    if isinstance(conv_module, nn.Conv2d):
        view = (conv_module.out_channels, 1, 1, 1)
    elif isinstance(conv_module, nn.ConvTranspose2d):
        view = (1, conv_module.out_channels, 1, 1)
    elif isinstance(conv_module, nn.Linear):
        view = (conv_module.out_features, 1)
    else:
        raise TypeError(f"Unsupported module type {type(conv_module)} in BN folding")
    if bn_module.affine:
        weight = w * (bn_module.weight / sigma).view(view)
        shift = bn_module.bias - bn_module.weight * bn_module.running_mean / sigma
        bias = shift if conv_module.bias is None else bn_module.weight * conv_module.bias / sigma + shift
    else:
        weight = w / sigma.view(view)
        shift = -bn_module.running_mean / sigma
        bias = shift if conv_module.bias is None else conv_module.bias / sigma + shift
    return weight, bias


def fold_bn_into_conv(conv_module, bn_module):
    w, b = _fold_bn(conv_module, bn_module)
    if conv_module.bias is None:
        conv_module.bias = nn.Parameter(b)
    else:
        conv_module.bias.data = b
    conv_module.weight.data = w
    # leave the (now unused) BN in a state that would be an identity, as the reference does
    bn_module.running_mean = bn_module.bias.data
    bn_module.running_var = bn_module.weight.data ** 2


def reset_bn(module: nn.BatchNorm2d):
    if module.track_running_stats:
        module.running_mean.zero_()
        module.running_var.fill_(1 - module.eps)
    if module.affine:
        nn.init.ones_(module.weight)
        nn.init.zeros_(module.bias)


def is_bn(m):
    return isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))


def is_absorbing(m):
    return isinstance(m, (nn.Conv2d, nn.Linear, nn.ConvTranspose2d))


def search_fold_and_remove_bn(model):
    """Depth-first: a BN that directly follows a conv / deconv / linear (in child order) is folded
    into it and replaced by ``StraightThrough``.  Children named in ``specials_unquantized_names``
    (e.g. ``codebook``) are left alone.  Returns the last absorbing layer seen."""
    model.eval()
    prev = None
    for name, child in model.named_children():
        if name in _skip_names():
            continue
        if is_bn(child) and is_absorbing(prev):
            fold_bn_into_conv(prev, child)
            setattr(model, name, StraightThrough())
        elif is_absorbing(child):
            prev = child
        else:
            prev = search_fold_and_remove_bn(child)
    return prev


def search_fold_and_reset_bn(model):
    model.eval()
    prev = None
    for _, child in model.named_children():
        if is_bn(child) and is_absorbing(prev):
            fold_bn_into_conv(prev, child)
        else:
            search_fold_and_reset_bn(child)
        prev = child
