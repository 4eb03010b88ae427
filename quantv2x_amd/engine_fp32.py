"""The UN-QUANTIZED model on the HIP path (SURVEY.md §8(b) "fp32 fall-backs for un-quantized mode").

``deploy(model)`` with a plain model -- what ``train_utils.create_model`` + ``load_saved_model`` hand the reference's
``opencood/tools/inference.py:106-170`` -- lands here: PillarVFE + scatter, BaseBEVBackbone and the shrinker run as f32-MFMA kernels
(``csrc/fp32_path.hip``: ``qv2x_pfn_scatter_f32``, ``qv2x_conv3x3_f32``, ``qv2x_deconv_f32``), the codebook encode takes fp32 rows
(``qv2x_codebook_encode_f32in``) and the decode + warp + attention kernel and the heads are the fp32 kernels the W8A8 path uses
(head output quantizers off).  BatchNorm is folded on the host in float64 (``fold_bn.py:19-127``'s algebra), which is the only
arithmetic difference to the reference's eval-mode forward besides summation order.

The state is plain numpy like the PTQ state: ``<module>/w`` (folded), ``<module>/bias``, ``codebook/...``, ``meta/...``, ``meta/mode = 'fp32'``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from .engine import DeployedModel, _dev, _pack_k4p, decode_tables
from .ptq_state import _np, _HEADS, extended_codebook


def _fold(w: np.ndarray, b: Optional[np.ndarray], bn, out_axis: int):
    """conv / linear weight + BatchNorm(eval) -> folded (w', b') in float64, returned as float32"""
    w = w.astype(np.float64)
    cout = w.shape[out_axis]
    b = np.zeros(cout) if b is None else b.astype(np.float64)
    if bn is None:
        return w.astype(np.float32), b.astype(np.float32)
    g, beta = _np(bn.weight).astype(np.float64), _np(bn.bias).astype(np.float64)
    mu, var = _np(bn.running_mean).astype(np.float64), _np(bn.running_var).astype(np.float64)
    s = g / np.sqrt(var + bn.eps)
    shape = [1] * w.ndim
    shape[out_axis] = cout
    return (w * s.reshape(shape)).astype(np.float32), (beta + (b - mu) * s).astype(np.float32)


def export_fp32_state(model) -> Dict[str, np.ndarray]:
    """Plain (un-quantized) ``HeterModelBaseline`` / ``HeterBaselineCollabCodebook`` (``_mc``) -> numpy state of the fp32 HIP path."""
    fusion = type(getattr(model, "fusion_net", None)).__name__
    if fusion not in ("AttFusion", "MaxFusion") or getattr(model, "shrink_flag", False) or getattr(model, "compress", False):
        raise NotImplementedError("deployed path: AttFusion or MaxFusion, no post-fusion shrink_conv, no compressor")
    out: Dict[str, np.ndarray] = {"meta/mode": np.array("fp32"), "meta/fusion_method": np.array("max" if fusion == "MaxFusion" else "att")}
    enc = model.encoder_m1
    vfe = enc.pillar_vfe
    if len(vfe.pfn_layers) != 1 or vfe.with_distance or not vfe.use_absolute_xyz:
        raise NotImplementedError("deployed PFN: one layer, use_absolute_xyz, no distance feature")
    pfn = vfe.pfn_layers[0]
    w, b = _fold(_np(pfn.linear.weight), None if pfn.linear.bias is None else _np(pfn.linear.bias), pfn.norm if pfn.use_norm else None, 0)
    if w.shape != (64, 10):
        raise NotImplementedError("deployed PFN expects Linear(10 -> 64)")
    out["pfn/w"], out["pfn/bias"] = w, b
    out["meta/voxel"] = np.array([vfe.voxel_x, vfe.voxel_y, vfe.voxel_z], dtype=np.float64)
    out["meta/offset"] = np.array([vfe.x_offset, vfe.y_offset, vfe.z_offset], dtype=np.float64)
    sc = enc.scatter
    out["meta/grid"] = np.array([sc.nx, sc.ny, sc.nz], dtype=np.int64)
    out["meta/HW_metres"] = np.array([model.H, model.W], dtype=np.float64)
    out["meta/discrete_ratio"] = np.float64(model.fake_voxel_size)
    bb = model.backbone_m1
    nums, strides, ups = [], [], []
    for lvl, blk in enumerate(bb.blocks):
        mods = list(blk)
        convs = [(mods[i], mods[i + 1]) for i in range(len(mods)) if isinstance(mods[i], nn.Conv2d)]
        nums.append(len(convs) - 1)
        strides.append(int(convs[0][0].stride[0]))
        for i, (cv, bn) in enumerate(convs):
            if cv.kernel_size != (3, 3) or not isinstance(bn, nn.BatchNorm2d):
                raise NotImplementedError("backbone blocks: 3x3 Conv2d + BatchNorm2d + ReLU")
            out[f"backbone_m1.blocks.{lvl}.{i + 1}/w"], out[f"backbone_m1.blocks.{lvl}.{i + 1}/bias"] = \
                _fold(_np(cv.weight), None if cv.bias is None else _np(cv.bias), bn, 0)
        de = list(bb.deblocks[lvl])
        if not isinstance(de[0], nn.ConvTranspose2d) or de[0].kernel_size != de[0].stride:
            raise NotImplementedError("deblocks: ConvTranspose2d with kernel == stride")
        ups.append(int(de[0].stride[0]))
        out[f"backbone_m1.deblocks.{lvl}.0/w"], out[f"backbone_m1.deblocks.{lvl}.0/bias"] = \
            _fold(_np(de[0].weight), None if de[0].bias is None else _np(de[0].bias), de[1], 1)
    out["meta/layer_nums"], out["meta/layer_strides"], out["meta/upsample_strides"] = (np.array(v, dtype=np.int64) for v in (nums, strides, ups))
    dc = model.shrinker_m1.layers[0].double_conv
    for i, cv in enumerate((dc[0], dc[2])):
        out[f"shrinker_m1.layers.0.double_conv.{i}/w"], out[f"shrinker_m1.layers.0.double_conv.{i}/bias"] = _fold(_np(cv.weight), _np(cv.bias), None, 0)
    out["meta/supervise_single"] = np.bool_(bool(getattr(model, "supervise_single", False)))
    for suffix in ("", "_single"):
        for h in _HEADS:
            m = getattr(model, h + suffix, None)
            if m is not None:
                out[f"{h}{suffix}/w"], out[f"{h}{suffix}/bias"] = _np(m.weight).astype(np.float32).reshape(m.weight.shape[0], -1), _np(m.bias).astype(np.float32)
    cb = getattr(model, "codebook", None)
    out["meta/has_codebook"] = np.bool_(cb is not None)
    if cb is not None:
        for lvl, (e, d) in enumerate(zip(cb._encoders, cb._decoders)):
            p = f"codebook/{lvl}/"
            out[p + "codebook"] = extended_codebook(_np(e._quantizer._codebook).astype(np.float32))   # [m * k, m * d] (ptq_state.py)
            for tag, lin in (("stage", e._latentStageEncoder), ("qhead", e._quantizationHead), ("lhead", e._latentHead),
                             ("dqhead", d._dequantizationHead), ("side", d._sideHead), ("restore", d._restoreHead)):
                if lin is not None:
                    out[p + tag + "_w"], out[p + tag + "_b"] = _np(lin.weight).astype(np.float32), _np(lin.bias).astype(np.float32)
        out["meta/codebook_levels"] = np.int64(len(cb._encoders))
        out["meta/codebook_segs"] = np.int64(cb._m)
    return out


def pack_k8(wmat: np.ndarray) -> np.ndarray:
    """[columns][K] -> [K/8][columns][2][4]: the load order of qv2x_conv3x3_f32 / qv2x_deconv_f32"""
    cols, k = wmat.shape
    assert k % 8 == 0
    return np.ascontiguousarray(wmat.reshape(cols, k // 8, 2, 4).transpose(1, 0, 2, 3), dtype=np.float32)


class _F32Conv:
    def __init__(self, state, name, stride, dev, c0=0):
        w = state[name + "/w"]                                           # [Cout][Cin][3][3]
        self.name, self.stride, self.cout, self.cin, self.c0 = name, stride, w.shape[0], w.shape[1], c0
        self.wmat = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(w.shape[0], -1))          # [Cout][tap * Cin + ci]
        self.w = _dev(pack_k8(self.wmat), dev)
        self.bias = _dev(state[name + "/bias"].astype(np.float32), dev)


class _F32Deconv:
    def __init__(self, state, name, dev):
        w = state[name + "/w"]                                           # [Cin][Cout][s][s]
        self.name, self.cin, self.cout, self.s = name, w.shape[0], w.shape[1], w.shape[2]
        self.wmat = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(-1, w.shape[0]))          # [(i*s + j)*Cout + co][ci]
        self.w = _dev(pack_k8(self.wmat), dev)
        self.bias = _dev(state[name + "/bias"].astype(np.float32), dev)


class _F32Heads:
    """cls | reg | dir stacked, output quantizers off (the same ``qv2x_heads_*`` kernels as the W8A8 path)"""

    def __init__(self, state, suffix, dev):
        ws = [state[h + suffix + "/w"] for h in _HEADS]
        bs = [state[h + suffix + "/bias"] for h in _HEADS]
        self.splits = [w.shape[0] for w in ws]
        self.cout = sum(self.splits)
        self.cout_pad = (self.cout + 31) // 32 * 32
        if self.cout_pad > 96:
            raise NotImplementedError("heads: at most 96 stacked output channels")
        pad = self.cout_pad - self.cout
        self.w = _dev(_pack_k4p(np.concatenate(ws + [np.zeros((pad, 256), np.float32)])), dev)
        self.bias = _dev(np.concatenate(bs + [np.zeros(pad, np.float32)]), dev)
        self.da = _dev(np.full(self.cout_pad, -1.0, np.float32), dev)
        self.za = _dev(np.zeros(self.cout_pad, np.float32), dev)


class DeployedFp32Model(DeployedModel):
    """Same call contract and the same tail (decode + warp + attention, heads, HIP-graph capture, multi-GPU stage interface) as
    ``DeployedModel``; a1-a6 in fp32."""

    def __init__(self, state: Dict[str, np.ndarray], device="cuda", emit_single_preds: Optional[bool] = None):
        nn.Module.__init__(self)
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.Qv2xError("DeployedFp32Model needs an MI355X (torch.cuda.is_available() is False)")
        self.state, self.dev = state, torch.device(device)
        s, dev = state, self.dev
        self.nx, self.ny, _ = (int(v) for v in s["meta/grid"])
        self.hm, self.wm = (float(v) for v in s["meta/HW_metres"])
        self.ratio = float(s["meta/discrete_ratio"])
        self.layer_nums = [int(v) for v in s["meta/layer_nums"]]
        self.strides = [int(v) for v in s["meta/layer_strides"]]
        self.ups = [int(v) for v in s["meta/upsample_strides"]]
        self.has_codebook = bool(s["meta/has_codebook"])
        self.compress = False
        self.fusion = 1 if str(s.get("meta/fusion_method", "att")) == "max" else 0
        self.emit_single = bool(s["meta/supervise_single"]) if emit_single_preds is None else bool(emit_single_preds)
        f32a = lambda a: (C.c_float * len(a))(*[float(np.float32(v)) for v in a])
        self.pfn_w, self.pfn_b = f32a(s["pfn/w"].reshape(-1)), f32a(s["pfn/bias"])
        self.pfn_vox, self.pfn_off = f32a(s["meta/voxel"]), f32a(s["meta/offset"])
        self.blocks: List[List[_F32Conv]] = []
        self.deblocks: List[_F32Deconv] = []
        for lvl in range(len(self.layer_nums)):
            self.blocks.append([_F32Conv(s, f"backbone_m1.blocks.{lvl}.{i + 1}", self.strides[lvl] if i == 0 else 1, dev)
                                for i in range(self.layer_nums[lvl] + 1)])
            self.deblocks.append(_F32Deconv(s, f"backbone_m1.deblocks.{lvl}.0", dev))
        self.cat_channels = sum(d.cout for d in self.deblocks)
        self.shrink0 = _F32Conv(s, "shrinker_m1.layers.0.double_conv.0", 1, dev)
        self.shrink1 = _F32Conv(s, "shrinker_m1.layers.0.double_conv.1", 1, dev)
        if self.shrink1.cout != 256:
            raise NotImplementedError("deployed path expects a 256-channel shared feature")
        if self.has_codebook:
            self.enc_levels, self.segs = int(s["meta/codebook_levels"]), int(s.get("meta/codebook_segs", 1))
            self.levels = self.enc_levels * self.segs                  # code planes (engine.py)
            self.ke = int(s["codebook/0/codebook"].shape[0])
            self.kc = self.ke // self.segs
            lut, lut_bias = decode_tables(s, self.enc_levels)
            lut = lut.reshape(self.levels, self.kc, lut.shape[-1])
            self.lut, self.lut_bias = _dev(lut, dev), _dev(lut_bias, dev)
            self.level_blobs = [self._level_blob(l) for l in range(self.enc_levels)]
            self.level_ptrs = (C.c_void_p * self.enc_levels)(*[b.data_ptr() for b in self.level_blobs])
        self.heads = _F32Heads(s, "", dev)
        self.heads_single = _F32Heads(s, "_single", dev) if (self.emit_single and "cls_head_single/w" in s) else None
        self._bufs: Dict[int, dict] = {}
        self.chains = [None] * len(self.blocks)
        self.use_wide_conv = self.batch_deconvs = self.use_chains = False

    # ---- buffers: fp32 NHWC with a zero border ---------------------------------------------------------------------------
    def _workspace(self, n: int) -> dict:
        if n in self._bufs:
            return self._bufs[n]
        z = lambda h, w, c: torch.zeros((n, h + 2, w + 2, c), dtype=torch.float32, device=self.dev)
        b = {"canvas": z(self.ny, self.nx, 64), "lvl": []}
        h, w = self.ny, self.nx
        for lvl, convs in enumerate(self.blocks):
            h, w = (h + 2 - 3) // self.strides[lvl] + 1, (w + 2 - 3) // self.strides[lvl] + 1
            b["lvl"].append(([z(h, w, convs[0].cout) for _ in range(2)], h, w))
        (_, h0, w0) = b["lvl"][0]
        self.fh, self.fw = h0 * self.ups[0], w0 * self.ups[0]
        for lvl, (_, hl, wl) in enumerate(b["lvl"]):
            if (hl * self.ups[lvl], wl * self.ups[lvl]) != (self.fh, self.fw):
                raise ValueError("the grid does not line up across backbone levels")
        b["cat"] = z(self.fh, self.fw, self.cat_channels)
        b["s0"], b["s1"] = z(self.fh, self.fw, self.shrink0.cout), z(self.fh, self.fw, self.shrink1.cout)
        if self.has_codebook:
            b["codes"] = torch.empty((self.levels, n, self.fh * self.fw), dtype=torch.uint8, device=self.dev)
        self._bufs[n] = b
        return b

    # ---- a1 - a6 -------------------------------------------------------------------------------------------------------------
    def pillars_to_canvas(self, inputs: dict, n_agents: int):
        b = self._workspace(n_agents)
        st = L.current_stream()
        vf = inputs["voxel_features"].contiguous()
        co = inputs["voxel_coords"].to(torch.int32).contiguous()
        npnt = inputs["voxel_num_points"].to(torch.int32).contiguous()
        if vf.dtype != torch.float32 or vf.dim() != 3 or tuple(vf.shape[1:]) != (32, 4):
            raise ValueError("voxel_features must be float32 [M, 32, 4]")
        canvas = b["canvas"]
        L.check(self.lib.qv2x_fill_i8(L.ptr(canvas), canvas.numel() * 4, 0, st), "qv2x_fill_i8")
        L.check(self.lib.qv2x_pfn_scatter_f32(L.ptr(vf), L.ptr(co), L.ptr(npnt), vf.shape[0], 32, self.pfn_w, self.pfn_b, self.pfn_vox,
                                              self.pfn_off, L.ptr(canvas), n_agents, self.ny, self.nx, st), "qv2x_pfn_scatter_f32")
        return canvas

    def _gemm(self, fn, layer, x, n, h, w, out, stride, cin0=0, out_c0=0):
        d = L.F32ConvDesc()
        d.n, d.h, d.w, d.cin_total, d.cin0, d.cin = n, h, w, x.shape[-1], cin0, layer.cin
        d.stride, d.cout, d.out_ctotal, d.out_c0, d.relu = stride, layer.cout, out.shape[-1], out_c0, 1
        L.check(fn(C.byref(d), L.ptr(x), L.ptr(layer.w), L.ptr(layer.bias), L.ptr(out), L.current_stream()), layer.name)

    def conv_plan(self, n_agents: int):
        b = self._workspace(n_agents)
        plan, x, h, w, c0 = [], b["canvas"], self.ny, self.nx, 0
        for lvl, convs in enumerate(self.blocks):
            pair, ho, wo = b["lvl"][lvl]
            for i, layer in enumerate(convs):
                plan.append(("conv", layer, x, h, w, pair[i % 2], 0, n_agents * ho * wo * layer.cout * 9 * layer.cin))
                x, h, w = pair[i % 2], ho, wo
            de = self.deblocks[lvl]
            plan.append(("deconv", de, x, h, w, b["cat"], c0, n_agents * h * w * de.cin * de.cout * de.s * de.s))
            c0 += de.cout
        hw = n_agents * self.fh * self.fw
        plan.append(("conv", self.shrink0, b["cat"], self.fh, self.fw, b["s0"], 0, hw * self.shrink0.cout * 9 * self.shrink0.cin))
        plan.append(("conv", self.shrink1, b["s0"], self.fh, self.fw, b["s1"], 0, hw * self.shrink1.cout * 9 * self.shrink1.cin))
        return plan

    def run_plan(self, n_agents: int, only=None, taps: Optional[dict] = None):
        for (kind, layer, x, h, w, out, c0, _) in self.conv_plan(n_agents):
            if only is not None and not only(kind, layer):
                continue
            if kind == "conv":
                self._gemm(self.lib.qv2x_conv3x3_f32, layer, x, n_agents, h, w, out, layer.stride)
            else:
                self._gemm(self.lib.qv2x_deconv_f32, layer, x, n_agents, h, w, out, layer.s, out_c0=c0)
            if taps is not None:
                taps[layer.name] = out.clone()

    def encode_codes(self, n_agents: int, out: Optional[torch.Tensor] = None):
        b = self._workspace(n_agents)
        codes = b["codes"] if out is None else out
        d = L.EncodeDesc()
        d.n, d.h, d.w, d.levels, d.kc, d.in_zx, d.in_delta, d.segs = n_agents, self.fh, self.fw, self.enc_levels, self.kc, 0, 1.0, self.segs
        L.check(self.lib.qv2x_codebook_encode_f32in(C.byref(d), L.ptr(b["s1"]), self.level_ptrs, L.ptr(codes), L.current_stream()),
                "qv2x_codebook_encode_f32in")
        return codes

    def encode_agents(self, inputs: dict, n_agents: int, taps: Optional[dict] = None):
        b = self._workspace(n_agents)
        canvas = self.pillars_to_canvas(inputs, n_agents)
        self.run_plan(n_agents, taps=taps)
        if taps is not None:
            taps["canvas"], taps["cat"] = canvas, b["cat"]
        return self.encode_codes(n_agents) if self.has_codebook else b["s1"]

    def _shared_features(self, shrinker_out, n_total: int):
        # a copy, not arithmetic: the interior of the padded fp32 tensor as contiguous rows
        return shrinker_out[:, 1:-1, 1:-1, :].reshape(n_total, self.fh * self.fw, 256).contiguous()
