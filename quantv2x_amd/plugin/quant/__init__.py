"""Mirror of the ``opencood.quant`` package surface used by ``opencood/tools/inference_quant.py``."""
from .adaptive_rounding import AdaRoundQuantizer
from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule, QuantSpconvModule, StraightThrough, UniformAffineQuantizer
from .quant_model import QuantModel
from .set_act_quantize_params import set_act_quantize_params
from .set_weight_quantize_params import save_quantized_weight, set_weight_quantize_params
from .block_recon import block_reconstruction, pyramid_reconstruction
from .encoder_recon import encoder_reconstruction
from .layer_recon import layer_reconstruction
