"""Frozen W8A8 state of a calibrated ``QuantModel`` as plain numpy arrays.

The reference never persists PTQ state -- calibration reruns on every ``inference_quant.py`` launch
(SURVEY.md §5 "Checkpoint / resume").  ``export_ptq_state`` walks a calibrated ``QuantModel`` -- the
reference's own (``opencood/quant/quant_model.py``) or this build's mirror; only attribute names are
used -- and collects exactly what the deployed integer path needs:

  <module>/w_code   uint8   clamp(round(w / delta_w) + zp_w, 0, 255)      (quant_layer.py:132-133)
                            or floor(w / delta_w) + (alpha >= 0) for AdaRound (adaptive_rounding.py:46-58)
  <module>/w_delta  f32 [dim0], <module>/w_zp f32 [dim0]   per dim-0 (C_out; C_in for ConvTranspose2d)
  <module>/bias     f32 [C_out]  (BN already folded, fold_bn.py)
  <module>/a_delta, <module>/a_zp   f32 scalars of the output activation quantizer
  pfn/a2_delta, pfn/a2_zp           the extra quantizer of QuantPFNLayer (quant_block.py:619-621)
  codebook/...                      fp32 heads + codebooks (the codebook is left unquantized)
  meta/...                          geometry of the BEV grid

``save_ptq_state`` / ``load_ptq_state`` store it as one ``.npz``.
"""
from __future__ import annotations

import re
from typing import Dict

import numpy as np
import torch

_HEADS = ("cls_head", "reg_head", "dir_head")


def _np(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()
    return np.asarray(t)


def _is_quant_module(m) -> bool:
    return all(hasattr(m, a) for a in ("weight_quantizer", "act_quantizer", "org_weight", "fwd_func"))


def weight_codes(qm) -> np.ndarray:
    """Integer weight codes of one ``QuantModule`` (uint8)."""
    wq = qm.weight_quantizer
    w = qm.weight.detach()
    delta, zp = torch.as_tensor(wq.delta), torch.as_tensor(wq.zero_point)
    if hasattr(wq, "alpha") and getattr(wq, "round_mode", "") == "learned_hard_sigmoid":
        code = torch.floor(w / delta) + (wq.alpha >= 0).float()
    else:
        code = torch.round(w / delta)
    return _np(torch.clamp(code + zp, 0, wq.n_levels - 1)).astype(np.uint8)


def _is_pyramid(model) -> bool:
    return hasattr(model, "pyramid_backbone")


def _residual_blocks(model):
    """(name, block) of every quantized residual block (QuantBasicBlock / QuantBottleneck, quant_block.py:68-131)."""
    return [(n, m) for n, m in model.named_modules() if type(m).__name__ in ("QuantBasicBlock", "QuantBottleneck")]


def _check_structure(model) -> None:
    """The deployed engines hard-wire two network shapes.  Anything else the plugin / the reference can
    build must be refused here, not silently run as a different network."""
    pyramid = _is_pyramid(model)
    if pyramid:
        pb = model.pyramid_backbone
        if type(pb).__name__ != "QuantPyramidFusion" or pb.stage != "collab" or not pb.model_cfg.get("resnext", False):
            raise NotImplementedError("deployed Pyramid path: a quantized PyramidFusion with resnext: true, stage: collab")
        if pb.align_corners:
            raise NotImplementedError("deployed Pyramid path: align_corners false")
        if not getattr(model, "shrink_flag", False):
            raise NotImplementedError("deployed Pyramid path: the post-fusion shrink_conv ('shrink_header') is part of the network")
        for m in getattr(model, "modality_name_list", ["m1"]):
            bb = getattr(model, "backbone_" + m)
            if len(bb.deblocks) != 0 or bb.num_levels != 1:
                raise NotImplementedError(f"deployed Pyramid path: a one-level per-agent ResNet backbone without deblocks (modality {m})")
        for m in getattr(model, "modality_name_list", ["m1"]):
            if type(getattr(model, "aligner_" + m).channel_align).__name__ != "Identity":
                raise NotImplementedError(f"deployed Pyramid path: aligner core_method identity (modality {m})")
    else:
        fusion = getattr(model, "fusion_net", None)
        if type(fusion).__name__ not in ("AttFusion", "MaxFusion"):
            raise NotImplementedError(f"deployed path: fusion_net must be AttFusion or MaxFusion (fusion_method 'att' / 'max'), got {type(fusion).__name__}")
        if getattr(model, "shrink_flag", False):
            raise NotImplementedError("deployed path: a post-fusion shrink_conv ('shrink_header' in the model args) is not built")
    if getattr(model, "compress", False) and (pyramid or getattr(model, "codebook", None) is not None):
        raise NotImplementedError("deployed path: the NaiveCompressor ('compressor' in the model args) is built for the baseline models "
                                  "without a codebook (the codebook models never call it, heter_baseline_collab_codebook.py:119-132)")
    for name, m in model.named_modules():
        if not _is_quant_module(m):
            continue
        if type(getattr(m, "norm_function", None)).__name__ != "StraightThrough":
            raise NotImplementedError(f"{name}: an unfolded norm_function ({type(m.norm_function).__name__}); build the QuantModel "
                                      "with is_fusing=True (BN folded into the convolution)")
        leaf = name.split(".")[-1]
        head = leaf.replace("_single", "") in _HEADS
        # the last convolution of a residual branch and the 1x1 shortcut feed the fp32 add (quant_block.py:76-84, :108-117)
        branch_end = pyramid and ".resnet." in name and (leaf in ("conv3", "downsample") or (leaf == "conv2" and name.startswith("backbone_m")))
        if bool(m.disable_act_quant) != branch_end and not head:
            raise NotImplementedError(f"{name}: disable_act_quant = {bool(m.disable_act_quant)} does not match the deployed network")
        act = type(getattr(m, "activation_function", None)).__name__
        bare = head or branch_end or name.endswith("pfn_layers.0.linear") or leaf.startswith("single_head_")
        if act != ("StraightThrough" if bare else "ReLU"):
            raise NotImplementedError(f"{name}: activation {act} does not match the deployed network (ReLU on every conv / deconv "
                                      "but the heads, the PFN linear and the ends of residual branches)")
    for name, b in _residual_blocks(model):
        if type(b.activation_function).__name__ != "ReLU" or not b.act_quantizer.inited or b.act_quantizer.n_bits != 8:
            raise NotImplementedError(f"{name}: residual blocks end in ReLU + a frozen 8-bit quantizer on the deployed path")


_PER_MODALITY = re.compile(r"^(encoder|backbone|shrinker|aligner)_(m\d+)(\.|$)")


def export_ptq_state(qt_model, modality: str = "m1") -> Dict[str, np.ndarray]:
    """``modality``: which of a heterogeneous model's per-modality stacks (``encoder_<m>`` / ``backbone_<m>`` / ``shrinker_<m>``,
    heter_model_baseline.py:41-75) this state carries, next to the shared codebook / fusion / heads.  Its keys are written under the
    ``_m1`` names, so every engine sees "its" modality as m1 (``engine.DeployedHeterModel`` holds one engine per modality)."""
    model = qt_model.model if hasattr(qt_model, "model") and not _is_quant_module(qt_model) else qt_model
    _check_structure(model)
    out: Dict[str, np.ndarray] = {}
    pyramid = _is_pyramid(model)
    if not hasattr(model, "encoder_" + modality):
        raise ValueError(f"the model has no modality {modality!r}")
    out["meta/fusion_method"] = np.array("pyramid" if pyramid else ("max" if type(getattr(model, "fusion_net", None)).__name__ == "MaxFusion" else "att"))
    if modality != "m1":
        out["meta/modality"] = np.array(modality)                   # (m1 states keep the key set the golden export pins)
    names, src_of = [], {}
    modules = dict(model.named_modules())
    for src_name, m in modules.items():
        if not _is_quant_module(m):
            continue
        pm = _PER_MODALITY.match(src_name)
        if pm and pm.group(2) != modality:
            continue                                                # another modality's stack: in that modality's own state
        name = _PER_MODALITY.sub(lambda g: f"{g.group(1)}_m1{g.group(3)}", src_name) if pm else src_name
        # W4A8 and the other sub-8-bit WEIGHT widths of the reference's PTQ script (scripts/inference/inference_quant.sh: --n_bits_w 4
        # --n_bits_a 8; quant_layer.py:337-340 bitwidth_refactor, quant_model.py:115-127 set_first_last_layer_to_8bit) deploy unchanged:
        # a b-bit code and its zero point lie in [0, 2^b - 1], stay uint8 and run on the same int8 kernels.  Activations stay 8-bit:
        # the requantizing epilogues clamp to [0, 255].
        if not 2 <= int(m.weight_quantizer.n_bits) <= 8:
            raise ValueError(f"{name}: weight bit width {m.weight_quantizer.n_bits} (2..8 deploy)")
        if m.act_quantizer.n_bits != 8:
            raise NotImplementedError(f"{name}: {m.act_quantizer.n_bits}-bit activations -- the deployed path is WxA8 (every epilogue clamps codes to [0, 255])")
        # AdaRoundQuantizer (adaptive_rounding.py:6-21) has no `inited`: it is built from an initialised quantizer
        if not (getattr(m.weight_quantizer, "inited", True) and m.act_quantizer.inited):
            raise ValueError(f"{name}: quantizers must be frozen (set_inited(True)) before export")
        names.append(name)
        src_of[name] = src_name
        wq, aq = m.weight_quantizer, m.act_quantizer
        out[name + "/w_code"] = weight_codes(m)
        out[name + "/w_delta"] = _np(torch.as_tensor(wq.delta)).reshape(-1).astype(np.float32)
        out[name + "/w_zp"] = _np(torch.as_tensor(wq.zero_point)).reshape(-1).astype(np.float32)
        out[name + "/bias"] = _np(m.bias).astype(np.float32) if m.bias is not None \
            else np.zeros(0, np.float32)
        out[name + "/a_delta"] = np.float32(_np(torch.as_tensor(aq.delta)).reshape(-1)[0])
        out[name + "/a_zp"] = np.float32(_np(torch.as_tensor(aq.zero_point)).reshape(-1)[0])
        out[name + "/a_off"] = np.bool_(bool(m.disable_act_quant))
    out["meta/module_names"] = np.array(names)
    out["meta/w_bits"] = np.array([int(modules[src_of[n]].weight_quantizer.n_bits) for n in names], dtype=np.int32)   # 8, or 2..7 (W4A8)

    enc = getattr(model, "encoder_" + modality)
    if type(enc).__name__ == "QuantSECOND":                       # the sparse encoder in front of the same 2-D path
        if pyramid:
            raise NotImplementedError("deployed Pyramid path: PointPillar agents")
        out.update(export_second_state(enc))
        shape = [int(v) for v in enc.spconv_block.sparse_shape]
        for _, m in second_layers(enc):
            shape = m.spconv_module.out_shape(shape)
        out["meta/encoder"] = np.array("second")
        out["meta/grid"] = np.array([shape[2], shape[1], shape[0]], dtype=np.int64)          # (W, H, D) of the BEV map the encoder hands over
        out["meta/canvas_channels"] = np.int64(enc.spconv_block.num_point_features * shape[0])
    else:
        vfe = enc.pillar_vfe
        if len(vfe.pfn_layers) != 1 or vfe.with_distance or not vfe.use_absolute_xyz:
            raise NotImplementedError("deployed PFN: one layer, use_absolute_xyz, no distance feature (the V2X-Real / OPV2V yaml)")
        pfn = vfe.pfn_layers[0]
        out["pfn/a2_delta"] = np.float32(_np(torch.as_tensor(pfn.act_quantizer.delta)).reshape(-1)[0])
        out["pfn/a2_zp"] = np.float32(_np(torch.as_tensor(pfn.act_quantizer.zero_point)).reshape(-1)[0])
        out["meta/voxel"] = np.array([vfe.voxel_x, vfe.voxel_y, vfe.voxel_z], dtype=np.float64)
        out["meta/offset"] = np.array([vfe.x_offset, vfe.y_offset, vfe.z_offset], dtype=np.float64)
        sc = enc.scatter
        out["meta/grid"] = np.array([sc.nx, sc.ny, sc.nz], dtype=np.int64)
        out["meta/encoder"] = np.array("point_pillar")
    out["meta/HW_metres"] = np.array([model.H, model.W], dtype=np.float64)
    out["meta/discrete_ratio"] = np.float64(model.fake_voxel_size)

    if pyramid:
        block_names = []
        for src_name, b in _residual_blocks(model):   # the quantizer after the fp32 add + ReLU (quant_block.py:92-96, :126-130)
            pm = _PER_MODALITY.match(src_name)
            if pm and pm.group(2) != modality:
                continue                              # (another modality's agent-side stack: HEAL's per-modality backbones)
            name = _PER_MODALITY.sub(lambda g: f"{g.group(1)}_m1{g.group(3)}", src_name) if pm else src_name
            out[name + "/a_delta"] = np.float32(_np(torch.as_tensor(b.act_quantizer.delta)).reshape(-1)[0])
            out[name + "/a_zp"] = np.float32(_np(torch.as_tensor(b.act_quantizer.zero_point)).reshape(-1)[0])
            block_names.append(name)
        out["meta/block_names"] = np.array(block_names)
        cfg_a, cfg_p = getattr(model, "backbone_" + modality).model_cfg, model.pyramid_backbone.model_cfg
        out["meta/layer_nums"] = np.array(cfg_a["layer_nums"], dtype=np.int64)
        out["meta/layer_strides"] = np.array(cfg_a["layer_strides"], dtype=np.int64)
        out["meta/pyramid_layer_nums"] = np.array(cfg_p["layer_nums"], dtype=np.int64)
        out["meta/pyramid_layer_strides"] = np.array(cfg_p["layer_strides"], dtype=np.int64)
        out["meta/upsample_strides"] = np.array(cfg_p["upsample_strides"], dtype=np.int64)
        out["meta/supervise_single"] = np.bool_(False)
    else:
        bb = getattr(model, "backbone_" + modality)
        out["meta/layer_nums"] = np.array([len(b) - 2 for b in bb.blocks], dtype=np.int64)
        out["meta/layer_strides"] = np.array([int(b[1].fwd_kwargs["stride"][0]) for b in bb.blocks], dtype=np.int64)
        out["meta/upsample_strides"] = np.array([int(d[0].fwd_kwargs["stride"][0]) for d in bb.deblocks], dtype=np.int64)
        out["meta/supervise_single"] = np.bool_(bool(getattr(model, "supervise_single", False)))

    out["meta/compress"] = np.bool_(bool(getattr(model, "compress", False)))
    cb = getattr(model, "codebook", None)
    out["meta/has_codebook"] = np.bool_(cb is not None)
    if cb is not None:
        m = int(cb._m)
        for lvl, (e, d) in enumerate(zip(cb._encoders, cb._decoders)):
            p = f"codebook/{lvl}/"
            out[p + "codebook"] = extended_codebook(_np(e._quantizer._codebook).astype(np.float32))    # [m * k, m * d]; m = 1: [k, d]
            for tag, lin in (("stage", e._latentStageEncoder), ("qhead", e._quantizationHead),
                             ("lhead", e._latentHead), ("dqhead", d._dequantizationHead),
                             ("side", d._sideHead), ("restore", d._restoreHead)):
                if lin is not None:
                    out[p + tag + "_w"] = _np(lin.weight).astype(np.float32)
                    out[p + tag + "_b"] = _np(lin.bias).astype(np.float32)
        out["meta/codebook_levels"] = np.int64(len(cb._encoders))
        out["meta/codebook_segs"] = np.int64(m)
        ks = {out[f"codebook/{lvl}/codebook"].shape[0] // m for lvl in range(len(cb._encoders))}
        if len(ks) != 1:
            raise NotImplementedError(f"deployed codebook path: one dict_size for every level (got {sorted(ks)})")
        k = ks.pop()
        if k > 256 or (m > 1 and k % 64) or (m == 1 and k % 32) or m not in (1, 2, 4) or m * k > 512:
            # (m k <= 512: the encode kernels' tile budget, qv2x_codebook_encode_f32 -- ADVICE r5: (4, 256) used to export and then fail
            #  with EINVAL on its first encode)
            raise NotImplementedError(f"deployed codebook path: seg_num 1 | 2 | 4 and dict_size <= 256 (a multiple of 32; of 64 with seg_num > 1; "
                                      f"seg_num * dict_size <= 512): got seg_num {m}, dict_size {k}")
    return check_finite(out)


def extended_codebook(cb: np.ndarray) -> np.ndarray:
    """``_multiCodebookQuantization._codebook`` f32 [m, k, d] (codebook.py:66-69) -> the EXTENDED form [m * k, m * d] the deployed path and
    the oracle work on: row s * k + j holds C[s][j] in dims [s d, (s + 1) d) and zeros elsewhere.  With it the per-segment arithmetic of
    codebook.py:115-131 / :192-201 is ordinary dense arithmetic, bit for bit: a row's dot product with q over all m * d dims is the
    segment's own ascending fma chain (fmaf(q, 0, acc) = acc), the decode head applied to a row is the head applied to the zero-padded
    codeword, and the wire's code plane (level l, segment s) indexes rows [s k, (s + 1) k) -- i.e. plane l * m + s of a table
    [levels * m][k][width].  m = 1: the plain [k, d] codebook."""
    m, k, d = cb.shape
    ext = np.zeros((m * k, m * d), np.float32)
    for s in range(m):
        ext[s * k:(s + 1) * k, s * d:(s + 1) * d] = cb[s]
    return ext


def check_finite(state: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Every float of a frozen PTQ state must be finite and every ``*_delta`` positive: the kernels' requantizer (csrc/common.h q_pack4)
    is specified on finite inputs only, and a finite state keeps every intermediate finite (exact i32 sums x finite scales + finite biases)."""
    for k, v in state.items():
        a = np.asarray(v)
        if a.dtype.kind == "f":
            # a DISABLED output quantizer's delta / zero point are never read by a kernel (e.g. an uninitialised delta of a layer with
            # disable_act_quant=True): exempt from both tests
            stem = k[:-len("a_delta")] if k.endswith("/a_delta") else (k[:-len("a_zp")] if k.endswith("/a_zp") else None)
            off = stem is not None and bool(state.get(stem + "a_off", False))
            if off:
                continue
            if not np.isfinite(a).all():
                raise ValueError(f"PTQ state: {k} holds a non-finite value; the deployed path is specified on finite quantizer parameters and weights")
            if k.endswith("_delta") and not (a > 0).all():
                raise ValueError(f"PTQ state: {k} must be positive")
    return state


def save_ptq_state(path: str, state: Dict[str, np.ndarray]) -> None:
    np.savez_compressed(path, **state)


def load_ptq_state(path: str) -> Dict[str, np.ndarray]:
    with np.load(path, allow_pickle=False) as z:
        return check_finite({k: z[k] for k in z.files})


def second_layers(enc):
    """The sparse convolutions of a ``QuantSECOND`` in execution order: (name, wrapper)."""
    block = enc.spconv_block
    out = []
    for stage in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
        for name, m in getattr(block, stage).named_modules():
            if type(m).__name__ == "QuantSpconvModule":
                out.append((f"spconv_block.{stage}.{name}", m))
    return out


def export_second_state(enc) -> Dict[str, np.ndarray]:
    """Frozen W8A8 state of a calibrated ``QuantSECOND`` (reference ``quant_block.py:1037-1078``).

    Per sparse convolution ``second/<i>/``: ``w_code`` u8 ``[K, C_in, C_out]`` (K = kz*ky*kx window offsets, z-major), ``w_delta`` /
    ``w_zp`` f32 ``[C_out]`` (the weight is ``[C_out, kz, ky, kx, C_in]``: dim 0 = C_out), ``bn_g`` / ``bn_h`` f32 ``[C_out]`` -- the
    BatchNorm1d that follows the accumulation, NOT folded (fold_bn.py only absorbs into Conv2d / Linear), as ``y = conv * g + h`` --
    ``a_delta`` / ``a_zp`` of the output quantizer, and ``geom`` i64 ``[10] = (subm, kz, ky, kx, sz, sy, sx, pz, py, px)``."""
    out: Dict[str, np.ndarray] = {}
    layers = second_layers(enc)
    for i, (name, m) in enumerate(layers):
        wq, aq, conv = m.weight_quantizer, m.act_quantizer, m.spconv_module
        if not 2 <= int(wq.n_bits) <= 8 or aq.n_bits != 8 or m.disable_act_quant:
            raise ValueError(f"{name}: the deployed path is WxA8 (2..8-bit weights, 8-bit activations) with every output quantized")
        if not (getattr(wq, "inited", True) and aq.inited):
            raise ValueError(f"{name}: quantizers must be frozen before export")
        if type(m.activation_function).__name__ != "ReLU" or type(m.norm_function).__name__ != "BatchNorm1d" or m.bias is not None:
            raise NotImplementedError(f"{name}: deployed sparse layers are conv (no bias) + BatchNorm1d + ReLU")
        code = weight_codes(m)                                              # [Cout, kz, ky, kx, Cin]
        co, kz, ky, kx, ci = code.shape
        p = f"second/{i}/"
        out[p + "w_code"] = np.ascontiguousarray(code.reshape(co, kz * ky * kx, ci).transpose(1, 2, 0))
        out[p + "w_delta"] = _np(torch.as_tensor(wq.delta)).reshape(-1).astype(np.float32)
        out[p + "w_zp"] = _np(torch.as_tensor(wq.zero_point)).reshape(-1).astype(np.float32)
        if out[p + "w_delta"].size != co:
            raise NotImplementedError(f"{name}: channel-wise weight quantizer expected")
        bn = m.norm_function
        g = (_np(bn.weight).astype(np.float64) / np.sqrt(_np(bn.running_var).astype(np.float64) + bn.eps))
        out[p + "bn_g"] = g.astype(np.float32)
        out[p + "bn_h"] = (_np(bn.bias).astype(np.float64) - _np(bn.running_mean).astype(np.float64) * g).astype(np.float32)
        out[p + "a_delta"] = np.float32(_np(torch.as_tensor(aq.delta)).reshape(-1)[0])
        out[p + "a_zp"] = np.float32(_np(torch.as_tensor(aq.zero_point)).reshape(-1)[0])
        out[p + "geom"] = np.array([int(conv.subm), *conv.kernel_size, *conv.stride, *conv.padding], dtype=np.int64)
    out["second/n_layers"] = np.int64(len(layers))
    out["second/sparse_shape"] = np.asarray(enc.spconv_block.sparse_shape, dtype=np.int64)      # (D, H, W) of the input volume
    out["second/names"] = np.array([n for n, _ in layers])
    return out
