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


def wrap(model, scale_method="minmax"):
    wq, aq = quant_params(scale_method=scale_method)
    qt = QuantModel(model, wq, aq).eval()
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
