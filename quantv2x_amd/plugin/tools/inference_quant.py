"""PTQ driver steps of the boundary; mirror of the flow in ``opencood/tools/inference_quant.py:225-322``:

    wq = {n_bits, channel_wise=True, scale_method}; aq = {n_bits, channel_wise=False, scale_method, leaf_param=True, prob}
    qt_model = QuantModel(model, wq, aq)              (BN folded, modules swapped in place)        :236-243
    set_weight_quantize_params(qt_model)                                                            :263
    [reconstruction | reset of the activation quantizers: inited=False -> EMA min-max on forward]   :58-64, 279-321
    qt_model.set_quant_state(True, True)                                                            :322

and then -- new in this build -- ``quantv2x_amd.deploy(qt_model)`` to run the frozen state on the HIP int8 path.
"""
import torch

from ..quant import QuantModel, UniformAffineQuantizer, set_weight_quantize_params


def quant_params(n_bits_w=8, n_bits_a=8, scale_method="minmax", prob=0.5):
    wq = dict(n_bits=n_bits_w, channel_wise=True, scale_method=scale_method)
    aq = dict(n_bits=n_bits_a, channel_wise=False, scale_method=scale_method, leaf_param=True, prob=prob)
    return wq, aq


def wrap(model, scale_method="minmax", n_bits_w=8, n_bits_a=8, first_last_8bit=False):
    """``first_last_8bit``: ``set_first_last_layer_to_8bit`` (quant_model.py:115-127) between the wrap and the weight-quantizer
    initialisation, where the driver has it (inference_quant.py:248-250) -- the usual companion of ``--n_bits_w 4``."""
    wq, aq = quant_params(n_bits_w, n_bits_a, scale_method=scale_method)
    qt = QuantModel(model, wq, aq).eval()
    if first_last_8bit:
        qt.set_first_last_layer_to_8bit()
    set_weight_quantize_params(qt)
    return qt


def activation_quantizers(qt):
    return [m for m in qt.modules() if isinstance(m, UniformAffineQuantizer) and m.leaf_param]


def calibrate_minmax(qt, batches, seed=0):
    """The no-reconstruction path of the reference driver: every activation quantizer observes ``batches`` with
    ``inited=False`` (EMA 0.9 / 0.1 over min-max ranges), then is frozen."""
    for a in activation_quantizers(qt):
        a.set_inited(False)
    qt.set_quant_state(True, True)
    with torch.no_grad():
        for dd in batches:
            torch.manual_seed(seed)      # the reference model's codebook forward draws Gumbel noise even in eval
            qt(dd)
    for a in activation_quantizers(qt):
        a.set_inited(True)
    return qt


# ---- reconstruction path of the driver (inference_quant.py:236-243, 268-322) ------------------------------------------------
def wrap_pair(model, scale_method="minmax", prob=0.5):
    """``(fp_model, qt_model)`` as the driver builds them: the fp twin keeps its BatchNorm layers (``is_fusing=False``) and runs
    with quantization off -- its BN statistics drive the distribution correction -- the quantized one is folded."""
    import copy
    wq, aq = quant_params(scale_method=scale_method, prob=prob)
    fp = QuantModel(copy.deepcopy(model), wq, aq, is_fusing=False).eval()
    fp.set_quant_state(False, False)
    qt = QuantModel(model, wq, aq).eval()
    set_weight_quantize_params(qt)
    return fp, qt


def recon_kwargs(cali_data, iters_w=5000, weight=0.01, b_start=20, b_end=2, warmup=0.2, lr=4e-5, input_prob=0.5, keep_cpu=False,
                 lamb_r=0.2, T=7.0, bn_lr=1e-3, lamb_c=0.02, **extra):
    """the ``kwargs`` dict of the driver (``:279-283``; defaults = its argparse defaults, ``:89-131``)"""
    return dict(cali_data=cali_data, iters=iters_w, weight=weight, b_range=(b_start, b_end), warmup=warmup, opt_mode='mse', lr=lr,
                input_prob=input_prob, keep_gpu=not keep_cpu, lamb_r=lamb_r, T=T, bn_lr=bn_lr, lamb_c=lamb_c, **extra)


def recon_model(qt_model, fp_model, kwargs, log=print):
    """Walk the two module trees in step and reconstruct every quantized unit in forward order (``recon_model``, ``:286-317``):
    a bare ``QuantModule`` -> layer, a backbone / shrinker / compressor block -> block, the pillar feature net -> encoder."""
    from ..quant.block_recon import block_reconstruction, pyramid_reconstruction
    from ..quant.encoder_recon import encoder_reconstruction
    from ..quant.layer_recon import layer_reconstruction
    from ..quant.quant_block import (QuantBaseBEVBackbone, QuantDownsampleConv, QuantNaiveCompressor, QuantPFNLayer, QuantPyramidFusion,
                                     QuantResNetBEVBackbone)
    from ..quant.quant_layer import QuantModule

    def walk(qt, fp):
        for (name, module), (_, fp_module) in zip(qt.named_children(), fp.named_children()):
            if isinstance(module, QuantModule):
                log('Reconstruction for layer {}'.format(name))
                layer_reconstruction(qt_model, fp_model, module, fp_module, **kwargs)
            elif isinstance(module, QuantPyramidFusion):          # a QuantResNetBEVBackbone subclass: test it first, as the reference does
                log('Reconstruction for pyramid fusion block {}'.format(name))
                pyramid_reconstruction(qt_model, fp_model, module, fp_module, **kwargs)
            elif isinstance(module, (QuantResNetBEVBackbone, QuantDownsampleConv, QuantBaseBEVBackbone, QuantNaiveCompressor)):
                log('Reconstruction for block {}'.format(name))
                block_reconstruction(qt_model, fp_model, module, fp_module, **kwargs)
            elif isinstance(module, QuantPFNLayer):
                log('Reconstruction for PointPillar PFN {}'.format(name))
                encoder_reconstruction(qt_model, fp_model, module, fp_module, **kwargs)
            else:
                walk(module, fp_module)
    walk(qt_model, fp_model)
    qt_model.set_quant_state(weight_quant=True, act_quant=True)
    qt_model.eval()
    return qt_model
