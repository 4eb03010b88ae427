"""``block_reconstruction`` for ``QuantBaseBEVBackbone`` / ``QuantDownsampleConv`` / ``QuantNaiveCompressor`` blocks; mirror of
``opencood/quant/block_recon.py:93-232``.  The shrinker block additionally matches the detection heads' output on the
reconstructed feature against the fp model's prediction (``forward_from_shrinker``, ``:76-91``, and the KL term ``:283-286``)."""
from .quant_block import BaseQuantBlock, QuantDownsampleConv
from .quant_model import QuantModel
from .recon import LinearTempDecay, LossFunction, forward_from_shrinker, reconstruct  # noqa: F401


def block_reconstruction(model: QuantModel, fp_model: QuantModel, block: BaseQuantBlock, fp_block: BaseQuantBlock, cali_data: list,
                         batch_size: int = 1, iters: int = 20000, weight: float = 0.01, opt_mode: str = 'mse', b_range: tuple = (20, 2),
                         warmup: float = 0.0, p: float = 2.0, lr: float = 4e-5, input_prob: float = 1.0, keep_gpu: bool = True,
                         lamb_r: float = 0.2, T: float = 7.0, bn_lr: float = 1e-3, lamb_c=0.02, **extra):
    reconstruct(model, fp_model, block, fp_block, cali_data, batch_size=batch_size, iters=iters, weight=weight, opt_mode=opt_mode,
                b_range=b_range, warmup=warmup, p=p, lr=lr, input_prob=input_prob, keep_gpu=keep_gpu, lamb_r=lamb_r, T=T, bn_lr=bn_lr,
                lamb_c=lamb_c, prediction_loss=isinstance(block, QuantDownsampleConv), **extra)


def pyramid_reconstruction(qt_model: QuantModel, fp_model: QuantModel, qt_block: BaseQuantBlock, fp_block: BaseQuantBlock, cali_data: list,
                           iters: int = 20000, weight: float = 0.01, opt_mode: str = 'mse', b_range: tuple = (20, 2), warmup: float = 0.0,
                           p: float = 2.0, lr: float = 4e-5, input_prob: float = 1.0, keep_gpu: bool = True, lamb_r: float = 0.2, T: float = 7.0,
                           bn_lr: float = 1e-3, lamb_c=0.02, **extra):
    """``QuantPyramidFusion`` as ONE reconstruction unit (``opencood/quant/pyramid_recon.py:124-278``): the fused feature against the fp
    twin's distribution-corrected one, the per-agent feature QDrop-mixed, ``record_len`` / ``affine_matrix`` / modality list passed
    through, prediction-level loss through ``shrink_conv`` and the heads (``forward_from_fusion :61-84``)."""
    from .recon import forward_from_fusion
    extra.pop("batch_size", None)
    reconstruct(qt_model, fp_model, qt_block, fp_block, cali_data, batch_size=1, iters=iters, weight=weight, opt_mode=opt_mode, b_range=b_range,
                warmup=warmup, p=p, lr=lr, input_prob=input_prob, keep_gpu=keep_gpu, lamb_r=lamb_r, T=T, bn_lr=bn_lr, lamb_c=lamb_c,
                prediction_loss=True, multi_input=True, prediction_fn=forward_from_fusion, **extra)
