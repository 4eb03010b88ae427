"""In-place graph surgery that turns a hot-path model into its fake-quant twin; mirror of
``opencood/quant/quant_model.py`` (``QuantModel :7-113``, helpers ``:115-147``).

Order of operations inside the constructor (``is_fusing=True``): fold BN -> replace registered block
types by their ``Quant*`` wrappers, bare ``Conv2d``/``Linear`` children by ``QuantModule`` and absorb a
following ReLU into the preceding ``QuantModule``.
"""
import torch.nn as nn

from .fold_bn import search_fold_and_remove_bn
from .quant_block import BaseQuantBlock, opencood_specials, specials, specials_unquantized_names  # noqa: F401
from .quant_layer import QuantModule, StraightThrough, UniformAffineQuantizer


class QuantModel(nn.Module):
    def __init__(self, model: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {},
                 is_fusing=True, skip_quant_module_names=None):
        super().__init__()
        self.skip_quant_module_names = tuple(skip_quant_module_names or [])
        if is_fusing:
            search_fold_and_remove_bn(model)
        self.model = model
        self._refactor(self.model, weight_quant_params, act_quant_params, "", absorb_bn=not is_fusing)

    def _should_skip_quantization(self, full_name: str, local_name: str) -> bool:
        return any(local_name == s or full_name == s or full_name.startswith(f"{s}.")
                   for s in self.skip_quant_module_names)

    def _refactor(self, module, wq, aq, parent_name, absorb_bn):
        last = None     # most recent QuantModule among this module's children
        for name, child in module.named_children():
            full = f"{parent_name}.{name}" if parent_name else name
            if name in specials_unquantized_names or self._should_skip_quantization(full, name):
                continue
            if type(child) in opencood_specials:
                setattr(module, name, opencood_specials[type(child)](child, wq, aq))
            elif isinstance(child, (nn.Conv2d, nn.Linear)):
                last = QuantModule(child, wq, aq)
                setattr(module, name, last)
            elif absorb_bn and isinstance(child, nn.BatchNorm2d):
                if last is not None:
                    last.norm_function = child
                    setattr(module, name, StraightThrough())
            elif isinstance(child, (nn.ReLU, nn.ReLU6)):
                if last is not None:
                    last.activation_function = child
                    setattr(module, name, StraightThrough())
            elif isinstance(child, StraightThrough):
                continue
            else:
                self._refactor(child, wq, aq, full, absorb_bn)

    # names kept for drivers that call them directly
    def quant_module_refactor(self, module, weight_quant_params={}, act_quant_params={}, parent_name=""):
        self._refactor(module, weight_quant_params, act_quant_params, parent_name, absorb_bn=False)

    def quant_module_refactor_wo_fuse(self, module, weight_quant_params={}, act_quant_params={}, parent_name=""):
        self._refactor(module, weight_quant_params, act_quant_params, parent_name, absorb_bn=True)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        for m in self.model.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.set_quant_state(weight_quant, act_quant)

    def forward(self, input):
        return self.model(input)

    def set_first_last_layer_to_8bit(self):
        weights, acts = [], []
        for m in self.model.modules():
            if isinstance(m, UniformAffineQuantizer):
                (acts if m.leaf_param else weights).append(m)
        weights[0].bitwidth_refactor(8)
        weights[-1].bitwidth_refactor(8)
        acts[-2].bitwidth_refactor(8)

    def disable_network_output_quantization(self):
        for name, m in self.model.named_modules():
            if isinstance(m, QuantModule) and name.rsplit(".", 1)[-1].startswith(("cls_head", "reg_head", "dir_head")):
                m.disable_act_quant = True

    def get_memory_footprint(self):
        total = sum(t.nelement() * t.element_size() for t in list(self.parameters()) + list(self.buffers()))
        return f"Model Memory Footprint: {total / (1024 ** 2):.2f} MB"
