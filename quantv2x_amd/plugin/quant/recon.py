"""AdaRound / QDrop reconstruction of one layer or block; the shared core behind ``layer_recon.layer_reconstruction``,
``block_recon.block_reconstruction`` and ``encoder_recon.encoder_reconstruction`` (reference: ``opencood/quant/
layer_recon.py:36-152``, ``block_recon.py:93-232``, ``encoder_recon.py:62-192`` -- three near-identical loops there, one here).

Per block:
  1. cache the quantized model's inputs to the block and the fp model's distribution-corrected targets (``data_utils``);
  2. initialise the block's activation quantizers on the cached inputs (``set_act_quantize_params``);
  3. swap every weight quantizer for an ``AdaRoundQuantizer`` in ``learned_hard_sigmoid`` mode (trainable ``alpha``), make every
     activation step size a trainable ``nn.Parameter`` (LSQ style) and switch QDrop on (``is_training``);
  4. ``iters`` Adam steps (alpha lr 3e-3; delta lr ``lr`` with cosine annealing) on
        Lp(block(drop_input), fp target)  +  rounding regulariser  [+ KL(pred || tgt) / lamb_r + Lp(prediction, fp prediction)]
     where ``drop_input`` takes each element from the quantized input with probability ``input_prob`` and from the corrected
     fp input otherwise, and the bracket is the prediction-level term of the shrinker block (``forward_from_shrinker``);
  5. freeze: hard rounding masks (``soft_targets = False``), QDrop off, ``trained = True``.

The frozen block is exactly what ``quantv2x_amd.ptq_state.export_ptq_state`` reads (``floor(w / delta) + (alpha >= 0)``,
``delta`` as a Parameter), so a model calibrated here on the GPU deploys on the HIP int8 path.
"""
import torch
import torch.nn.functional as F

from .adaptive_rounding import AdaRoundQuantizer
from .data_utils import first_output, get_dc_fp_init, get_init, save_block_extras
from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule, lp_loss
from .set_act_quantize_params import set_act_quantize_params


class LinearTempDecay:
    """temperature ``b`` of the rounding regulariser: ``start_b`` until ``rel_start_decay * t_max``, then linear to ``end_b``"""

    def __init__(self, t_max: int, rel_start_decay: float = 0.2, start_b: int = 10, end_b: int = 2):
        self.t_max, self.start_decay, self.start_b, self.end_b = t_max, rel_start_decay * t_max, start_b, end_b

    def __call__(self, t):
        if t < self.start_decay:
            return self.start_b
        rel_t = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + (self.start_b - self.end_b) * max(0.0, (1 - rel_t))


class LossFunction:
    """reconstruction + rounding (+ prediction-level) loss; ``count`` advances once per call"""

    def __init__(self, block, round_loss: str = 'relaxation', weight: float = 1., rec_loss: str = 'mse', max_count: int = 2000,
                 b_range: tuple = (10, 2), decay_start: float = 0.0, warmup: float = 0.0, p: float = 2., lam: float = 1.0, T: float = 7.0,
                 verbose: bool = True):
        if rec_loss != 'mse':
            raise ValueError('Not supported reconstruction loss function: {}'.format(rec_loss))
        if round_loss not in ('relaxation', 'none'):
            raise NotImplementedError
        self.block, self.round_loss, self.weight, self.p, self.lam, self.T = block, round_loss, weight, p, lam, T
        self.loss_start = max_count * warmup
        self.temp_decay = LinearTempDecay(max_count, rel_start_decay=warmup + (1 - warmup) * decay_start, start_b=b_range[0], end_b=b_range[1])
        self.count = 0
        self.verbose = verbose
        self.hetero_loss = torch.nn.KLDivLoss(reduction='batchmean')

    def __call__(self, pred, tgt, output_qt=None, output_fp=None):
        self.count += 1
        rec = lp_loss(pred, tgt, p=self.p)
        hetero = misalign = 0
        if output_qt is not None and output_fp is not None:
            hetero = self.hetero_loss(F.log_softmax(pred / self.T, dim=1), F.softmax(tgt / self.T, dim=1)) / self.lam
            misalign = lp_loss(output_qt, output_fp, p=self.p)
        b = self.temp_decay(self.count)
        rounding = 0
        if self.count < self.loss_start or self.round_loss == 'none':
            b = 0
        else:
            for m in self.block.modules():
                if isinstance(m, QuantModule):
                    vals = m.weight_quantizer.get_soft_targets()
                    rounding = rounding + self.weight * (1 - ((vals - .5).abs() * 2).pow(b)).sum()
        total = rec + rounding + (hetero + misalign if output_qt is not None and output_fp is not None else 0)
        if self.verbose and self.count % 200 == 0:
            print(f"[Iter {self.count}] total {float(total):.5f} | rec {float(rec):.5f} | round {float(rounding):.5f} | "
                  f"KL {float(hetero):.5f} | pred {float(misalign):.5f} | b {b:.3f}")
        return total


def forward_from_shrinker(model, feature: torch.Tensor):
    """the detection heads on a shrinker output (the rest of the baseline model minus fusion), ``block_recon.py:76-91``"""
    parts = []
    for name in ("cls_head", "reg_head", "dir_head"):
        head = getattr(model, name, None)
        if head is None:
            continue
        w = getattr(head, "weight", None)
        if isinstance(w, torch.Tensor) and w.dim() >= 2 and feature.dim() >= 2 and w.shape[1] != feature.shape[1]:
            continue
        parts.append(head(feature))
    if not parts:
        return None
    return torch.cat(parts, dim=1) if len(parts) >= 2 else parts[0]


def forward_from_fusion(model, fused_feature: torch.Tensor):
    """the rest of the Pyramid model after ``pyramid_backbone``: ``shrink_conv`` and the heads (``pyramid_recon.py:61-84``)"""
    if getattr(model, "shrink_flag", False):
        fused_feature = model.shrink_conv(fused_feature)
    return forward_from_shrinker(model, fused_feature)


def _pick(store, idx):
    return store[idx] if isinstance(store, torch.Tensor) else store[int(idx)]


def reconstruct(model, fp_model, block, fp_block, cali_data: list, batch_size: int = 1, iters: int = 20000, weight: float = 0.01,
                opt_mode: str = 'mse', b_range: tuple = (20, 2), warmup: float = 0.0, p: float = 2.0, lr: float = 4e-5,
                input_prob: float = 1.0, keep_gpu: bool = True, lamb_r: float = 0.2, T: float = 7.0, bn_lr: float = 1e-3, lamb_c=0.02,
                prediction_loss: bool = False, dc_iters: int = 500, verbose: bool = True, seed=None, multi_input: bool = False,
                prediction_fn=None):
    """``multi_input``: the block takes further positional inputs that are passed through unchanged (``QuantPyramidFusion``: record_len,
    affine_matrix, agent_modality_list, cam_crop_info) and may return a tuple whose first element is reconstructed;
    ``prediction_fn(model, block_output)``: the model's tail for the prediction-level loss (default ``forward_from_shrinker``)."""
    device = next(model.parameters()).device
    gen = torch.Generator(device="cpu")
    if seed is not None:
        gen.manual_seed(seed)
    # 1. calibration captures
    cached_inps = get_init(model, block, cali_data, batch_size=batch_size, input_prob=True, keep_gpu=keep_gpu)
    cached_outs, cached_output, cur_syms = get_dc_fp_init(fp_model, fp_block, cali_data, batch_size=batch_size, input_prob=True,
                                                         keep_gpu=keep_gpu, bn_lr=bn_lr, lamb=lamb_c, dc_iters=dc_iters)
    sz = cached_inps.size(0) if isinstance(cached_inps, torch.Tensor) else len(cached_inps)
    extras = save_block_extras(model, block, cali_data) if multi_input else None
    prediction_fn = prediction_fn or forward_from_shrinker
    # 2. activation quantizers
    set_act_quantize_params(block, cached_inps=cached_inps[:min(256, sz)], extras=extras)
    block.set_quant_state(True, True)
    for para in model.parameters():
        para.requires_grad = False
    # 3. trainable rounding + step sizes
    w_para, a_para = [], []
    holders = [m for m in block.modules() if isinstance(m, (QuantModule, BaseQuantBlock))]
    for m in holders:
        if isinstance(m, QuantModule):
            m.weight_quantizer = AdaRoundQuantizer(uaq=m.weight_quantizer, round_mode='learned_hard_sigmoid',
                                                   weight_tensor=m.org_weight.data.to(device))
            m.weight_quantizer.soft_targets = True
            w_para.append(m.weight_quantizer.alpha)
        if hasattr(m, 'act_quantizer'):
            if m.act_quantizer.delta is not None:
                m.act_quantizer.delta = torch.nn.Parameter(torch.as_tensor(m.act_quantizer.delta).detach().clone().to(device))
                a_para.append(m.act_quantizer.delta)
            m.act_quantizer.is_training = True
    w_opt = torch.optim.Adam(w_para, lr=3e-3) if w_para else None
    a_opt = torch.optim.Adam(a_para, lr=lr) if a_para else None
    a_sched = torch.optim.lr_scheduler.CosineAnnealingLR(a_opt, T_max=iters, eta_min=0.) if a_opt else None
    loss_func = LossFunction(block, round_loss='relaxation', weight=weight, max_count=iters, rec_loss=opt_mode, b_range=b_range,
                             decay_start=0, warmup=warmup, p=p, lam=lamb_r, T=T, verbose=verbose)
    # 4. optimisation
    for _ in range(iters):
        idx = int(torch.randint(0, sz, (), generator=gen))
        cur_inp = _pick(cached_inps, idx).to(device)
        cur_sym = _pick(cur_syms, idx).to(device)
        cur_out = _pick(cached_outs, idx).to(device)
        output_fp = _pick(cached_output, idx).to(device)
        drop_inp = torch.where(torch.rand(cur_inp.shape, generator=gen).to(device) < input_prob, cur_inp, cur_sym) if input_prob < 1.0 else cur_inp
        if w_opt:
            w_opt.zero_grad()
        if a_opt:
            a_opt.zero_grad()
        out_drop = first_output(block(drop_inp, *(extras[idx] if extras is not None else ())))
        output_qt = None
        if prediction_loss:
            output_qt = prediction_fn(model.model, out_drop)  # None when no head takes this feature width: the reference then skips the
            if output_qt is not None:                         # prediction-level terms (block_recon.py:190-200)
                want = output_fp if output_fp.dim() == output_qt.dim() else output_fp.unsqueeze(0)
                if output_qt.shape == want.shape:
                    output_fp = want
                else:
                    output_qt = None                            # e.g. fused prediction [B, ...] vs per-agent features [N, ...]
        err = loss_func(out_drop, cur_out, output_qt, output_fp if output_qt is not None else None)
        err.backward()
        if w_opt:
            w_opt.step()
        if a_opt:
            a_opt.step()
        if a_sched:
            a_sched.step()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    # 5. freeze
    for m in holders:
        if isinstance(m, QuantModule):
            m.weight_quantizer.soft_targets = False
        m.trained = True
        if hasattr(m, 'act_quantizer'):
            m.act_quantizer.is_training = False
    for m in fp_block.modules():
        if isinstance(m, (QuantModule, BaseQuantBlock)):
            m.trained = True
