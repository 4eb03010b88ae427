"""``encoder_reconstruction`` for the pillar feature net (``QuantPFNLayer``: pillar tensors ``[M, 32, 10]`` whose M changes from
batch to batch, BatchNorm1d -- no BatchNorm2d statistics to correct towards); mirror of ``opencood/quant/encoder_recon.py:62-192``."""
from .quant_block import BaseQuantBlock
from .quant_model import QuantModel
from .recon import LinearTempDecay, LossFunction, reconstruct  # noqa: F401


def encoder_reconstruction(qt_model: QuantModel, fp_model: QuantModel, qt_block: BaseQuantBlock, fp_block: BaseQuantBlock, cali_data: list,
                           batch_size: int = 1, iters: int = 20000, weight: float = 0.01, opt_mode: str = 'mse', b_range: tuple = (20, 2),
                           warmup: float = 0.0, p: float = 2.0, lr: float = 4e-5, input_prob: float = 1.0, keep_gpu: bool = True,
                           lamb_r: float = 0.2, T: float = 7.0, bn_lr: float = 1e-3, lamb_c=0.02, **extra):
    reconstruct(qt_model, fp_model, qt_block, fp_block, cali_data, batch_size=batch_size, iters=iters, weight=weight, opt_mode=opt_mode,
                b_range=b_range, warmup=warmup, p=p, lr=lr, input_prob=input_prob, keep_gpu=keep_gpu, lamb_r=lamb_r, T=T, bn_lr=bn_lr,
                lamb_c=lamb_c, prediction_loss=False, **extra)
