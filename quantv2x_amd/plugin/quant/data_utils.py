"""Calibration data capture for the reconstruction loops; mirror of ``opencood/quant/data_utils.py``
(``save_inp_oup_data :46-92``, ``GetLayerInpOut :137-164``, ``save_dc_fp_data :14-43``, ``GetDcFpLayerInpOut :166-286``) and of
``extract_prediction_tensor`` (``encoder_recon_utils.py:25-53``).

Two captures per block and calibration batch:

* the QUANTIZED model's input to the block (forward hook + ``StopForwardException``: the forward is abandoned at the block);
* the FP model's block output after *distribution correction* (DC): the block's fp input is nudged by ``dc_iters`` Adam steps
  (500 in the reference) so that the inputs of the BatchNorm2d layers inside the block match their running statistics, held
  near the original input by ``lp_loss / lamb``; the block output at that corrected input is the reconstruction target, the
  corrected input is what QDrop mixes into the quantized input, and the fp model's prediction tensor is kept for the
  prediction-level loss of the shrinker block.

The reference hard-codes ``.cuda()``; here every tensor follows the model's device, so the same code calibrates on an MI355X
(PyTorch-ROCm: autograd is needed, the HIP int8 kernels only run the frozen result) and runs in the CPU test-suite.
"""
from typing import Union

import torch
import torch.nn as nn
import torch.optim as optim

from ..tools import train_utils
from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule, lp_loss
from .quant_model import QuantModel


class StopForwardException(Exception):
    """Raised by a forward hook to abandon the rest of a forward pass."""


class DataSaverHook:
    """Forward hook that keeps a module's input and / or output, optionally stopping the forward there."""

    def __init__(self, store_input=False, store_output=False, stop_forward=False):
        self.store_input, self.store_output, self.stop_forward = store_input, store_output, stop_forward
        self.input_store = None
        self.output_store = None

    def __call__(self, module, input_batch, output_batch):
        if self.store_input:
            self.input_store = input_batch
        if self.store_output:
            self.output_store = output_batch
        if self.stop_forward:
            raise StopForwardException


def extract_prediction_tensor(output):
    """The model-level prediction tensor cached for the prediction loss: ``preds_tensor``, else cls | reg | dir concatenated,
    else the first nested dict that has one."""
    if not isinstance(output, dict):
        return None
    if isinstance(output.get("preds_tensor"), torch.Tensor):
        return output["preds_tensor"]
    parts = [output[k] for k in ("cls_preds", "reg_preds", "dir_preds") if isinstance(output.get(k), torch.Tensor)]
    if parts:
        return torch.cat(parts, dim=1)
    for v in output.values():
        if isinstance(v, dict):
            t = extract_prediction_tensor(v)
            if t is not None:
                return t
    return None


def _device_of(model):
    return next(model.parameters()).device


class GetLayerInpOut:
    """input of ``layer`` inside the quantized ``model`` for one calibration batch (tensor, dict, or None)"""

    def __init__(self, model: QuantModel, layer: Union[QuantModule, BaseQuantBlock], device, input_prob: bool = False):
        self.model, self.layer, self.device, self.input_prob = model, layer, device, input_prob
        self.data_saver = DataSaverHook(store_input=True, store_output=False, stop_forward=True)

    def __call__(self, model_input):
        handle = self.layer.register_forward_hook(self.data_saver)
        try:
            with torch.no_grad():
                self.model.set_quant_state(weight_quant=True, act_quant=True)
                try:
                    self.model(train_utils.to_device(model_input, self.device))
                except StopForwardException:
                    pass
        finally:
            handle.remove()
        kept = self.data_saver.input_store
        if kept is None or len(kept) == 0:
            return None
        self.extras = tuple(kept[1:])        # a multi-input block (PyramidFusion: record_len, affine_matrix, modalities, crop info)
        return kept[0] if isinstance(kept[0], dict) else kept[0].detach()


def first_output(out):
    """the tensor a block is reconstructed on: ``PyramidFusion`` returns ``(fused_feature, occupancy maps)``"""
    return out[0] if isinstance(out, (tuple, list)) else out


def save_block_extras(model: QuantModel, block, cali_data: list):
    """per calibration batch, the positional inputs of ``block`` after the first (``pyramid_recon_utils.get_pyramid_input``)"""
    grab = GetLayerInpOut(model, block, device=_device_of(model))
    out = []
    for batch in cali_data:
        if grab(batch) is not None:
            out.append(grab.extras)
    return out


def save_inp_oup_data(model: QuantModel, block, cali_data: list, batch_size: int = 1, keep_gpu: bool = True, input_prob: bool = False):
    """Inputs of ``block`` over the calibration set: one stacked tensor ``[N, ...]`` when every batch gives the same leading
    size, else the list of per-batch inputs (dict inputs, or pillar tensors ``[M, P, C]`` whose M varies)."""
    device = _device_of(model)
    grab = GetLayerInpOut(model, block, device=device, input_prob=input_prob)
    kept, lead, any_dict = [], [], False
    for batch in cali_data:
        x = grab(batch)
        if x is None or not isinstance(x, (torch.Tensor, dict)):
            continue
        if isinstance(x, dict):
            kept.append(x)
            any_dict = True
        else:
            lead.append(x.size(0))
            kept.append(x.unsqueeze(0).cpu())
    if not kept:
        raise RuntimeError("No valid calibration inputs for this layer; check calibration data.")
    if any_dict or len(set(lead)) != 1:
        return [x if isinstance(x, dict) else x[0] for x in kept]      # per-batch inputs as the block receives them
    stacked = torch.cat(kept)
    return stacked.to(device) if keep_gpu else stacked


class GetDcFpLayerInpOut:
    """fp block output at the distribution-corrected input, the fp model's prediction tensor, and the corrected input"""

    def __init__(self, model: QuantModel, layer, device, input_prob: bool = False, lamb=50, bn_lr=1e-3, dc_iters: int = 500):
        self.model, self.layer, self.device, self.input_prob = model, layer, device, input_prob
        self.lamb, self.bn_lr, self.dc_iters, self.eps = lamb, bn_lr, dc_iters, 1e-6
        self.data_saver = DataSaverHook(store_input=True, store_output=True, stop_forward=False)
        self.bns = [m for m in layer.modules() if isinstance(m, nn.BatchNorm2d)]
        self.bn_stats = [(m.running_mean.detach().clone().flatten().to(device),
                          torch.sqrt(m.running_var + self.eps).detach().clone().flatten().to(device)) for m in self.bns]

    @staticmethod
    def own_loss(a, b):
        return (a - b).norm() ** 2 / b.size(0)

    def __call__(self, model_input):
        self.model.set_quant_state(False, False)
        handle = self.layer.register_forward_hook(self.data_saver)
        try:
            with torch.no_grad():
                output_fp = extract_prediction_tensor(self.model(train_utils.to_device(model_input, self.device)))
        finally:
            handle.remove()
        kept = self.data_saver.input_store
        if output_fp is None or kept is None or len(kept) == 0:
            return None
        input_sym = kept[0] if isinstance(kept[0], dict) else kept[0].detach()
        extras = tuple(kept[1:])
        if isinstance(input_sym, dict):
            raise NotImplementedError("distribution correction of a dict-valued block input")
        para_input = input_sym.data.clone().to(self.device)
        para_input.requires_grad = True
        seen = {}
        hooks = [bn.register_forward_hook(lambda m, i, o, k=k: seen.__setitem__(k, i)) for k, bn in enumerate(self.bns)]
        opt = optim.Adam([para_input], lr=self.bn_lr)
        sched = optim.lr_scheduler.ReduceLROnPlateau(opt, min_lr=1e-5, patience=100)
        try:
            for _ in range(self.dc_iters):
                self.layer.zero_grad()
                opt.zero_grad()
                seen.clear()
                self.layer(para_input, *extras)
                mean_loss, std_loss = 0, 0
                for k, (bn_mean, bn_std) in enumerate(self.bn_stats):
                    inp = seen.get(k)
                    if not inp or inp[0] is None:
                        continue
                    flat = inp[0].view(inp[0].size(0), inp[0].size(1), -1)
                    mean_loss = mean_loss + self.own_loss(bn_mean, torch.mean(flat, dim=2))
                    std_loss = std_loss + self.own_loss(bn_std, torch.sqrt(torch.var(flat, dim=2) + self.eps))
                total = mean_loss + std_loss + lp_loss(para_input, input_sym) / self.lamb
                total.backward()
                opt.step()
                sched.step(total.item())
        finally:
            for h in hooks:
                h.remove()
        with torch.no_grad():
            out_fp = first_output(self.layer(para_input, *extras))
        out_fp = out_fp.unsqueeze(0)
        if self.input_prob:
            return out_fp.detach(), output_fp.detach(), para_input.unsqueeze(0).detach()
        return out_fp.detach(), output_fp.detach()


def save_dc_fp_data(model: QuantModel, layer, cali_data: list, batch_size: int = 32, keep_gpu: bool = True, input_prob: bool = False,
                    lamb=50, bn_lr=1e-3, dc_iters: int = 500):
    """Corrected fp targets over the calibration set.  Stacked tensors when the batches agree in shape, lists otherwise (the
    reference concatenates unconditionally and therefore only handles fixed-size block inputs; the pillar encoder's
    ``[M, P, C]`` inputs go through ``encoder_recon_utils`` there -- one code path here)."""
    device = _device_of(model)
    grab = GetDcFpLayerInpOut(model, layer, device=device, input_prob=input_prob, lamb=lamb, bn_lr=bn_lr, dc_iters=dc_iters)
    rows = []
    for batch in cali_data:
        r = grab(batch)
        if r is not None:
            rows.append(tuple(t.cpu() for t in r))
    if not rows:
        raise RuntimeError("No valid calibration targets for this layer; check calibration data.")
    cols = list(zip(*rows))
    same = all(len({tuple(t.shape) for t in col}) == 1 for col in cols)
    if same:
        cols = [torch.cat(col) for col in cols]
        if keep_gpu:
            cols = [c.to(device) for c in cols]
    else:
        cols = [[t[0] for t in col] for col in cols]                   # per-batch tensors without the stacking axis
    return tuple(cols)


def get_init(model, block, cali_data, batch_size, input_prob: bool = False, keep_gpu: bool = True):
    """``set_weight_quantize_params.get_init`` (``:5-7``)"""
    return save_inp_oup_data(model, block, cali_data, batch_size, input_prob=input_prob, keep_gpu=keep_gpu)


def get_dc_fp_init(model, block, cali_data, batch_size, input_prob: bool = False, keep_gpu: bool = True, lamb=50, bn_lr=1e-3, dc_iters: int = 500):
    """``set_weight_quantize_params.get_dc_fp_init`` (``:9-11``)"""
    return save_dc_fp_data(model, block, cali_data, batch_size, input_prob=input_prob, keep_gpu=keep_gpu, lamb=lamb, bn_lr=bn_lr, dc_iters=dc_iters)
