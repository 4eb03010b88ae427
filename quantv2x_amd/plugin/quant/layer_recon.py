"""``layer_reconstruction`` for a single ``QuantModule`` (the 1x1 heads); mirror of ``opencood/quant/layer_recon.py:36-152``."""
from .quant_layer import QuantModule
from .quant_model import QuantModel
from .recon import LinearTempDecay, LossFunction, reconstruct  # noqa: F401  (re-exported like the reference module)


def layer_reconstruction(model: QuantModel, fp_model: QuantModel, layer: QuantModule, fp_layer: QuantModule, cali_data: list,
                         batch_size: int = 1, iters: int = 20000, weight: float = 0.001, opt_mode: str = 'mse', b_range: tuple = (20, 2),
                         warmup: float = 0.0, p: float = 2.0, lr: float = 4e-5, input_prob: float = 1.0, keep_gpu: bool = True,
                         lamb_r: float = 0.2, T: float = 7.0, bn_lr: float = 1e-3, lamb_c=0.02, **extra):
    reconstruct(model, fp_model, layer, fp_layer, cali_data, batch_size=batch_size, iters=iters, weight=weight, opt_mode=opt_mode,
                b_range=b_range, warmup=warmup, p=p, lr=lr, input_prob=input_prob, keep_gpu=keep_gpu, lamb_r=lamb_r, T=T, bn_lr=bn_lr,
                lamb_c=lamb_c, prediction_loss=False, **extra)
