"""Weight-quantizer initialisation; mirror of ``opencood/quant/set_weight_quantize_params.py:13-24``."""
from .quant_layer import QuantModule, QuantSpconvModule


def set_weight_quantize_params(model):
    """Run every weight quantizer once with ``inited=False`` so it derives (delta, zero_point) from the weight."""
    for m in model.modules():
        if isinstance(m, (QuantSpconvModule, QuantModule)):
            m.weight_quantizer.set_inited(False)
            m.weight_quantizer(m.weight)
            m.weight_quantizer.set_inited(True)


def save_quantized_weight(model):
    for m in model.modules():
        if isinstance(m, (QuantSpconvModule, QuantModule)):
            m.weight.data = m.weight_quantizer(m.weight)
