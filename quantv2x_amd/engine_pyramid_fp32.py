"""The UN-QUANTIZED HEAL Pyramid model on the HIP path: what ``create_model`` + ``load_saved_model`` hand the reference's plain
``opencood/tools/inference.py`` / ``inference_mc_codebook_encdec.py`` for ``heter_pyramid_collab[_codebook]_mc[_encdec]``.

Every convolution is an f32-MFMA GEMM of ``csrc/fp32_path.hip`` (BatchNorm folded on the host in float64): 3x3 convolutions and deblocks
as in ``engine_fp32.py``; a 1x1 convolution is the deconvolution kernel with stride 1; the strided 1x1 shortcut is the 3x3 kernel with
only the centre tap non-zero; a grouped 3x3 (ResNeXt 32 x 4d) is one dense launch per 64-channel slab (block-diagonal weights through
the kernels' channel windows); ``relu(branch + shortcut)`` is ``qv2x_add_relu_f32``.  Maps are padded fp32 NHWC with a zero border.
The codebook runs ``qv2x_codebook_encode64_f32in`` / ``qv2x_codebook_decode_f32``; occupancy -> score is ``qv2x_occ_sigmoid_f32``,
the fusion ``qv2x_pyramid_weighted_fuse_f32p``; the heads are the fp32 heads kernel with the output quantizers off."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from .engine import _dev, decode_tables
from .engine_fp32 import _F32Heads, _fold, pack_k8
from .ptq_state import _HEADS, _np, extended_codebook


def export_fp32_pyramid_state(model) -> Dict[str, np.ndarray]:
    """Plain ``HeterPyramidCollabMC`` / ``HeterPyramidCollabCodebookMC[EncDec]`` -> numpy state of the fp32 Pyramid engine."""
    pb = getattr(model, "pyramid_backbone", None)
    if type(pb).__name__ != "PyramidFusion" or pb.stage != "collab" or not pb.model_cfg.get("resnext", False) or pb.align_corners:
        raise NotImplementedError("deployed Pyramid path: PyramidFusion with resnext: true, stage: collab, align_corners false")
    if getattr(model, "compress", False) or not getattr(model, "shrink_flag", False) or len(model.backbone_m1.deblocks) != 0:
        raise NotImplementedError("deployed Pyramid path: no compressor, a post-fusion shrink_conv, an agent backbone without deblocks")
    out: Dict[str, np.ndarray] = {"meta/mode": np.array("fp32"), "meta/fusion_method": np.array("pyramid")}
    enc = model.encoder_m1
    vfe = enc.pillar_vfe
    if len(vfe.pfn_layers) != 1 or vfe.with_distance or not vfe.use_absolute_xyz:
        raise NotImplementedError("deployed PFN: one layer, use_absolute_xyz, no distance feature")
    pfn = vfe.pfn_layers[0]
    out["pfn/w"], out["pfn/bias"] = _fold(_np(pfn.linear.weight), None if pfn.linear.bias is None else _np(pfn.linear.bias), pfn.norm if pfn.use_norm else None, 0)
    out["meta/voxel"] = np.array([vfe.voxel_x, vfe.voxel_y, vfe.voxel_z], dtype=np.float64)
    out["meta/offset"] = np.array([vfe.x_offset, vfe.y_offset, vfe.z_offset], dtype=np.float64)
    out["meta/grid"] = np.array([enc.scatter.nx, enc.scatter.ny, enc.scatter.nz], dtype=np.int64)
    out["meta/HW_metres"] = np.array([model.H, model.W], dtype=np.float64)
    out["meta/discrete_ratio"] = np.float64(model.fake_voxel_size)

    def conv(name, cv, bn):
        out[name + "/w"], out[name + "/bias"] = _fold(_np(cv.weight), None if cv.bias is None else _np(cv.bias), bn, 0)

    def blocks(prefix, resnet, names):
        for li in range(resnet.layernum):
            for bi, blk in enumerate(getattr(resnet, f"layer{li}")):
                base = f"{prefix}.layer{li}.{bi}"
                for i, n in enumerate(names):
                    conv(f"{base}.{n}", getattr(blk, n), getattr(blk, f"bn{i + 1}"))
                if blk.downsample is not None:
                    conv(base + ".downsample", blk.downsample[0], blk.downsample[1])
    blocks("backbone_m1.resnet", model.backbone_m1.resnet, ("conv1", "conv2"))
    blocks("pyramid_backbone.resnet", pb.resnet, ("conv1", "conv2", "conv3"))
    for lvl in range(pb.num_levels):
        head = getattr(pb, f"single_head_{lvl}")
        out[f"pyramid_backbone.single_head_{lvl}/w"], out[f"pyramid_backbone.single_head_{lvl}/bias"] = _np(head.weight).astype(np.float32), _np(head.bias).astype(np.float32)
        de = list(pb.deblocks[lvl])
        out[f"pyramid_backbone.deblocks.{lvl}.0/w"], out[f"pyramid_backbone.deblocks.{lvl}.0/bias"] = \
            _fold(_np(de[0].weight), None if de[0].bias is None else _np(de[0].bias), de[1], 1)
    cfg_a, cfg_p = model.backbone_m1.model_cfg, pb.model_cfg
    for k, v in (("meta/layer_nums", cfg_a["layer_nums"]), ("meta/layer_strides", cfg_a["layer_strides"]), ("meta/pyramid_layer_nums", cfg_p["layer_nums"]),
                 ("meta/pyramid_layer_strides", cfg_p["layer_strides"]), ("meta/upsample_strides", cfg_p["upsample_strides"])):
        out[k] = np.array(v, dtype=np.int64)
    dc = model.shrink_conv.layers[0].double_conv
    for i, cv in enumerate((dc[0], dc[2])):
        conv(f"shrink_conv.layers.0.double_conv.{i}", cv, None)
    for h in _HEADS:
        m = getattr(model, h)
        out[f"{h}/w"], out[f"{h}/bias"] = _np(m.weight).astype(np.float32).reshape(m.weight.shape[0], -1), _np(m.bias).astype(np.float32)
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


# ---- GEMM views of the layers: [columns][K] row-major, K = tap * cin + ci for the 3x3 kernel, = ci for the 1x1 / deconv kernel ---------
def dense3x3(w: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(w.shape[0], -1))


def centre_tap3x3(w: np.ndarray) -> np.ndarray:
    """a 1x1 (strided) convolution as the 3x3 kernel sees it: only tap (1, 1) carries the weights"""
    full = np.zeros((w.shape[0], 9, w.shape[1]), np.float32)
    full[:, 4, :] = w.reshape(w.shape[0], w.shape[1])
    return full.reshape(w.shape[0], -1)


def grouped_slabs(w: np.ndarray) -> List[np.ndarray]:
    """[c, cg, 3, 3] grouped weights -> per 64-channel slab the dense [64][9 * 64] block-diagonal matrix"""
    c, cg = w.shape[:2]
    slabs = []
    for s0 in range(0, c, 64):
        m = np.zeros((64, 9, 64), np.float32)
        for co in range(64):
            g0 = (co // cg) * cg
            m[co, :, g0:g0 + cg] = w[s0 + co].reshape(cg, 9).T
        slabs.append(m.reshape(64, -1))
    return slabs


class _G:
    """one launch of the fp32 GEMM kernel: packed weights + bias + the shape bits the descriptor needs"""

    def __init__(self, name, wmat, bias, dev, cin, cout, stride=1, deconv=False, relu=True, cin0=0, out_c0=0):
        self.name, self.wmat, self.bias_np = name, wmat, bias
        self.w, self.bias = _dev(pack_k8(wmat), dev), _dev(bias.astype(np.float32), dev)
        self.cin, self.cout, self.stride, self.deconv, self.relu, self.cin0, self.out_c0 = cin, cout, stride, deconv, relu, cin0, out_c0


class DeployedPyramidFp32Model(nn.Module):
    """Same call contract as ``DeployedPyramidModel`` (``forward`` = the hard encode -> decode path, ``encode_features`` / ``decode_features``)."""

    def __init__(self, state: Dict[str, np.ndarray], device="cuda"):
        super().__init__()
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.Qv2xError("DeployedPyramidFp32Model needs an MI355X (torch.cuda.is_available() is False)")
        self.state, self.dev = state, torch.device(device)
        s, dev = state, self.dev
        self.nx, self.ny, _ = (int(v) for v in s["meta/grid"])
        self.hm, self.wm = (float(v) for v in s["meta/HW_metres"])
        self.ratio = float(s["meta/discrete_ratio"])
        self.has_codebook = bool(s["meta/has_codebook"])
        f32a = lambda a: (C.c_float * len(a))(*[float(np.float32(v)) for v in a])
        self.pfn_w, self.pfn_b = f32a(s["pfn/w"].reshape(-1)), f32a(s["pfn/bias"])
        self.pfn_vox, self.pfn_off = f32a(s["meta/voxel"]), f32a(s["meta/offset"])
        self.agent_stride = int(s["meta/layer_strides"][0])
        self.p_strides = [int(v) for v in s["meta/pyramid_layer_strides"]]
        self.ups = [int(v) for v in s["meta/upsample_strides"]]
        g = lambda n: (s[n + "/w"], s[n + "/bias"])
        self.agent_blocks = []
        for b in range(int(s["meta/layer_nums"][0])):
            base = f"backbone_m1.resnet.layer0.{b}"
            st = self.agent_stride if b == 0 else 1
            blk = {"name": base, "stride": st}
            w, bi = g(base + ".conv1"); blk["conv1"] = _G(base + ".conv1", dense3x3(w), bi, dev, w.shape[1], w.shape[0], st)
            w, bi = g(base + ".conv2"); blk["conv2"] = _G(base + ".conv2", dense3x3(w), bi, dev, w.shape[1], w.shape[0], 1, relu=False)
            if base + ".downsample/w" in s:
                w, bi = g(base + ".downsample"); blk["down"] = _G(base + ".downsample", centre_tap3x3(w), bi, dev, w.shape[1], w.shape[0], st, relu=False)
            self.agent_blocks.append(blk)
        self.D = self.agent_blocks[-1]["conv2"].cout
        if self.has_codebook:
            self.enc_levels, self.segs = int(s["meta/codebook_levels"]), int(s.get("meta/codebook_segs", 1))
            self.levels = self.enc_levels * self.segs                  # code planes (engine.py)
            self.ke, d = (int(v) for v in s["codebook/0/codebook"].shape)
            self.kc = self.ke // self.segs
            if d != 64 or self.D != 64:
                raise NotImplementedError("deployed fp32 Pyramid codebook: the 64-wide one")
            lut, lut_bias = decode_tables(s, self.enc_levels, 64)
            lut = lut.reshape(self.levels, self.kc, 64)
            self.lut, self.lut_bias = _dev(lut, dev), _dev(lut_bias, dev)
            from .engine_pyramid import DeployedPyramidModel
            self.native64 = True
            self.level_blobs = [DeployedPyramidModel._level_blob(self, l) for l in range(self.enc_levels)]
            self.level_ptrs = (C.c_void_p * self.enc_levels)(*[b.data_ptr() for b in self.level_blobs])
        self.pyr_blocks, self.occ, self.deblocks = [], [], []
        for lvl, nb in enumerate(int(v) for v in s["meta/pyramid_layer_nums"]):
            blocks = []
            for b in range(nb):
                base = f"pyramid_backbone.resnet.layer{lvl}.{b}"
                st = self.p_strides[lvl] if b == 0 else 1
                blk = {"name": base, "stride": st}
                w, bi = g(base + ".conv1"); blk["conv1"] = _G(base + ".conv1", w.reshape(w.shape[0], -1), bi, dev, w.shape[1], w.shape[0], 1, deconv=True)
                w, bi = g(base + ".conv2")
                blk["conv2"] = [_G(f"{base}.conv2[{i}]", m, bi[64 * i:64 * i + 64], dev, 64, 64, st, cin0=64 * i, out_c0=64 * i) for i, m in enumerate(grouped_slabs(w))]
                blk["width"] = w.shape[0]
                w, bi = g(base + ".conv3"); blk["conv3"] = _G(base + ".conv3", w.reshape(w.shape[0], -1), bi, dev, w.shape[1], w.shape[0], 1, deconv=True, relu=False)
                blk["planes"] = w.shape[0]
                if base + ".downsample/w" in s:
                    w, bi = g(base + ".downsample"); blk["down"] = _G(base + ".downsample", centre_tap3x3(w), bi, dev, w.shape[1], w.shape[0], st, relu=False)
                blocks.append(blk)
            self.pyr_blocks.append(blocks)
            w, bi = g(f"pyramid_backbone.single_head_{lvl}")
            wm = np.zeros((64, w.shape[1]), np.float32); wm[0] = w.reshape(-1)                    # one real column, padded to the kernel's 64
            bm = np.zeros(64, np.float32); bm[0] = bi[0]
            self.occ.append(_G(f"pyramid_backbone.single_head_{lvl}", wm, bm, dev, w.shape[1], 64, 1, deconv=True, relu=False))
            w, bi = g(f"pyramid_backbone.deblocks.{lvl}.0")
            self.deblocks.append(_G(f"pyramid_backbone.deblocks.{lvl}.0", np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(-1, w.shape[0])), bi, dev,
                                    w.shape[0], w.shape[1], int(w.shape[2]), deconv=True))
        self.cat_channels = sum(d.cout for d in self.deblocks)
        w, bi = g("shrink_conv.layers.0.double_conv.0"); self.shrink0 = _G("shrink_conv.layers.0.double_conv.0", dense3x3(w), bi, dev, w.shape[1], w.shape[0])
        w, bi = g("shrink_conv.layers.0.double_conv.1"); self.shrink1 = _G("shrink_conv.layers.0.double_conv.1", dense3x3(w), bi, dev, w.shape[1], w.shape[0])
        self.heads = _F32Heads(s, "", dev)
        self._bufs: Dict[tuple, dict] = {}

    # ---- buffers: fp32 NHWC with a zero border --------------------------------------------------------------------------------------
    def _z(self, n, h, w, c):
        return torch.zeros((n, h + 2, w + 2, c), dtype=torch.float32, device=self.dev)

    def _ws(self, n: int, nb: int) -> dict:
        key = (n, nb)
        if key in self._bufs:
            return self._bufs[key]
        h, w = self.ny, self.nx
        self.fh, self.fw = (h - 1) // self.agent_stride + 1, (w - 1) // self.agent_stride + 1
        fh, fw = self.fh, self.fw
        b = {"canvas": self._z(n, h, w, 64), "c1": self._z(n, fh, fw, 64), "y": self._z(n, fh, fw, 64), "ds": self._z(n, fh, fw, 64),
             "x": [self._z(n, fh, fw, 64) for _ in range(2)], "feat": self._z(n, fh, fw, 64), "lvl": []}
        if self.has_codebook:
            b["codes"] = torch.empty((self.levels, n, fh * fw), dtype=torch.uint8, device=self.dev)
            b["rows"] = torch.empty((n * fh * fw, 64), dtype=torch.float32, device=self.dev)
        for lvl, blocks in enumerate(self.pyr_blocks):
            hi, wi = h, w = (fh, fw) if lvl == 0 else (h, w)
            st = self.p_strides[lvl]
            h, w = (hi - 1) // st + 1, (wi - 1) // st + 1
            width, planes = blocks[0]["width"], blocks[0]["planes"]
            b["lvl"].append({"h": h, "w": w, "t1_in": self._z(n, hi, wi, width), "t1": self._z(n, h, w, width), "t2": self._z(n, h, w, width),
                             "y": self._z(n, h, w, planes), "ds": self._z(n, h, w, planes), "x": [self._z(n, h, w, planes) for _ in range(2)],
                             "occ64": self._z(n, h, w, 64), "occ": torch.empty((n, h * w), dtype=torch.float32, device=self.dev),
                             "score": torch.empty((n, h * w), dtype=torch.float32, device=self.dev), "fused": self._z(nb, h, w, planes)})
        oh, ow = b["lvl"][0]["h"] * self.ups[0], b["lvl"][0]["w"] * self.ups[0]
        for lvl, lv in enumerate(b["lvl"]):
            if (lv["h"] * self.ups[lvl], lv["w"] * self.ups[lvl]) != (oh, ow):
                raise ValueError("the grid does not line up across pyramid levels")
        self.oh, self.ow = oh, ow
        b["cat"], b["s0"], b["s1"] = self._z(nb, oh, ow, self.cat_channels), self._z(nb, oh, ow, self.shrink0.cout), self._z(nb, oh, ow, self.shrink1.cout)
        self._bufs[key] = b
        return b

    # ---- launches --------------------------------------------------------------------------------------------------------------------
    def _gemm(self, g: _G, x, n, h, w, out):
        d = L.F32ConvDesc()
        d.n, d.h, d.w, d.cin_total, d.cin0, d.cin = n, h, w, x.shape[-1], g.cin0, g.cin
        d.stride, d.cout, d.out_ctotal, d.out_c0, d.relu = g.stride, g.cout, out.shape[-1], g.out_c0, 1 if g.relu else 0
        fn = self.lib.qv2x_deconv_f32 if g.deconv else self.lib.qv2x_conv3x3_f32
        L.check(fn(C.byref(d), L.ptr(x), L.ptr(g.w), L.ptr(g.bias), L.ptr(out), L.current_stream()), g.name)

    def _add_relu(self, a, b, out):
        L.check(self.lib.qv2x_add_relu_f32(L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.current_stream()), "qv2x_add_relu_f32")

    # ---- the agent side ----------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def agent_features(self, inputs: dict, n: int, nb: int = 1, taps: Optional[dict] = None):
        b = self._ws(n, nb)
        st = L.current_stream()
        vf = inputs["voxel_features"].contiguous()
        co = inputs["voxel_coords"].to(torch.int32).contiguous()
        npnt = inputs["voxel_num_points"].to(torch.int32).contiguous()
        if vf.dtype != torch.float32 or vf.dim() != 3 or tuple(vf.shape[1:]) != (32, 4):
            raise ValueError("voxel_features must be float32 [M, 32, 4]")
        canvas = b["canvas"]
        L.check(self.lib.qv2x_fill_i8(L.ptr(canvas), canvas.numel() * 4, 0, st), "qv2x_fill_i8")
        L.check(self.lib.qv2x_pfn_scatter_f32(L.ptr(vf), L.ptr(co), L.ptr(npnt), vf.shape[0], 32, self.pfn_w, self.pfn_b, self.pfn_vox, self.pfn_off,
                                              L.ptr(canvas), n, self.ny, self.nx, st), "qv2x_pfn_scatter_f32")
        x, h, w = canvas, self.ny, self.nx
        for i, blk in enumerate(self.agent_blocks):
            out = b["feat"] if i == len(self.agent_blocks) - 1 else b["x"][i % 2]
            self._gemm(blk["conv1"], x, n, h, w, b["c1"])
            self._gemm(blk["conv2"], b["c1"], n, self.fh, self.fw, b["y"])
            if "down" in blk:
                self._gemm(blk["down"], x, n, h, w, b["ds"])
                self._add_relu(b["y"], b["ds"], out)
            else:
                self._add_relu(b["y"], x, out)
            if taps is not None:
                taps[blk["name"]] = out.clone()
            x, h, w = out, self.fh, self.fw
        return x

    @torch.no_grad()
    def encode_features(self, inputs: dict, n: int, taps: Optional[dict] = None, out: Optional[torch.Tensor] = None):
        if not self.has_codebook:
            raise L.Qv2xError("encode_features: this model has no codebook (no wire format); call forward")
        feat = self.agent_features(inputs, n, 1, taps)
        codes = self._ws(n, 1)["codes"] if out is None else out
        d = L.EncodeDesc()
        d.n, d.h, d.w, d.levels, d.kc, d.in_zx, d.in_delta, d.segs = n, self.fh, self.fw, self.enc_levels, self.kc, 0, 1.0, self.segs
        L.check(self.lib.qv2x_codebook_encode64_f32in(C.byref(d), 64, L.ptr(feat), self.level_ptrs, L.ptr(codes), L.current_stream()), "qv2x_codebook_encode64_f32in")
        return codes

    # ---- the ego side ------------------------------------------------------------------------------------------------------------------
    def _pyramid_and_heads(self, x, lens: List[int], pairwise: torch.Tensor, ego: int, taps: Optional[dict]) -> dict:
        n, nb = sum(lens), len(lens)
        b = self._ws(n, nb)
        h, w = self.fh, self.fw
        occ_maps, c0 = [], 0
        for lvl, blocks in enumerate(self.pyr_blocks):
            lv = b["lvl"][lvl]
            for i, blk in enumerate(blocks):
                out = lv["x"][i % 2]
                ho, wo = lv["h"], lv["w"]
                t1 = lv["t1_in"] if (h, w) != (ho, wo) else lv["t1"]
                self._gemm(blk["conv1"], x, n, h, w, t1)
                for slab in blk["conv2"]:
                    self._gemm(slab, t1, n, h, w, lv["t2"])
                self._gemm(blk["conv3"], lv["t2"], n, ho, wo, lv["y"])
                if "down" in blk:
                    self._gemm(blk["down"], x, n, h, w, lv["ds"])
                    self._add_relu(lv["y"], lv["ds"], out)
                else:
                    self._add_relu(lv["y"], x, out)
                if taps is not None:
                    taps[blk["name"]] = out.clone()
                x, h, w = out, ho, wo
            self._gemm(self.occ[lvl], x, n, h, w, lv["occ64"])
            L.check(self.lib.qv2x_occ_sigmoid_f32(L.ptr(lv["occ64"]), n, h, w, 64, L.ptr(lv["occ"]), L.ptr(lv["score"]), L.current_stream()), "qv2x_occ_sigmoid_f32")
            occ_maps.append(lv["occ"].view(n, 1, h, w).clone())
            start = 0
            for bi, na in enumerate(lens):
                fd = L.FuseDesc()
                fd.agents, fd.h, fd.w, fd.levels, fd.kc, fd.max_cav, fd.ego = na, h, w, 1, 1, pairwise.shape[1], ego
                fd.h_metres, fd.w_metres, fd.discrete_ratio = self.hm, self.wm, self.ratio
                L.check(self.lib.qv2x_pyramid_weighted_fuse_f32p(C.byref(fd), blocks[-1]["planes"], L.ptr(x[start:start + na]), L.ptr(lv["score"][start:start + na]),
                                                                 L.ptr(pairwise[bi]), L.ptr(lv["fused"][bi]), L.current_stream()), "qv2x_pyramid_weighted_fuse_f32p")
                start += na
            de = self.deblocks[lvl]
            de.out_c0 = c0
            self._gemm(de, lv["fused"], nb, h, w, b["cat"])
            c0 += de.cout
            if taps is not None:
                taps[f"score{lvl}"], taps[f"fused{lvl}"] = lv["score"].clone(), lv["fused"].clone()
        self._gemm(self.shrink0, b["cat"], nb, self.oh, self.ow, b["s0"])
        self._gemm(self.shrink1, b["s0"], nb, self.oh, self.ow, b["s1"])
        rows = b["s1"][:, 1:-1, 1:-1, :].reshape(nb * self.oh * self.ow, 256).contiguous()       # a copy, not arithmetic
        hd = self.heads
        preds = torch.empty((nb, hd.cout, self.oh, self.ow), dtype=torch.float32, device=self.dev)
        L.check(self.lib.qv2x_heads_f32(L.ptr(rows), nb * self.oh * self.ow, self.oh * self.ow, hd.cout, hd.cout_pad, L.ptr(hd.w), L.ptr(hd.bias),
                                        L.ptr(hd.da), L.ptr(hd.za), L.ptr(preds), L.current_stream()), "qv2x_heads_f32")
        if taps is not None:
            taps["cat"], taps[self.shrink1.name] = b["cat"], b["s1"]
        c, r, _ = hd.splits
        return {"pyramid": "collab", "cls_preds": preds[:, :c], "reg_preds": preds[:, c:c + r], "dir_preds": preds[:, c + r:], "preds_tensor": preds,
                "occ_single_list": occ_maps}

    @torch.no_grad()
    def decode_features(self, codes, agent_stride: int, level_stride: int, lens: List[int], pairwise: torch.Tensor, ego: int = 0,
                        taps: Optional[dict] = None) -> dict:
        if not self.has_codebook:
            raise L.Qv2xError("decode_features: this model has no codebook (no wire format); call forward")
        n, nb = sum(lens), len(lens)
        b = self._ws(n, nb)
        hw = self.fh * self.fw
        cptr = codes if isinstance(codes, C.c_void_p) else L.ptr(codes)
        L.check(self.lib.qv2x_codebook_decode_f32(cptr, agent_stride, level_stride, n, hw, self.levels, self.kc, 64, L.ptr(self.lut), L.ptr(self.lut_bias),
                                                  L.ptr(b["rows"]), L.current_stream()), "qv2x_codebook_decode_f32")
        b["feat"][:, 1:-1, 1:-1, :].copy_(b["rows"].view(n, self.fh, self.fw, 64))               # rows -> the padded map (a copy)
        if taps is not None:
            taps["features"] = b["rows"]
        return self._pyramid_and_heads(b["feat"], lens, pairwise, ego, taps)

    @torch.no_grad()
    def forward(self, data_dict: dict, taps: Optional[dict] = None) -> dict:
        agents = data_dict["agent_modality_list"]
        n_total = len(agents)
        if any(str(a) != "m1" for a in agents):
            raise NotImplementedError("deployed path: every agent is the LiDAR modality 'm1'")
        pairwise = data_dict["pairwise_t_matrix"]
        if pairwise.dtype != torch.float64 or not pairwise.is_contiguous():
            pairwise = pairwise.to(torch.float64).contiguous()
        if pairwise.shape[0] == 1:
            lens = [n_total]
        else:
            rl = data_dict["record_len"]
            lens = [int(v) for v in (rl.tolist() if isinstance(rl, torch.Tensor) else rl)]
        if not self.has_codebook:
            x = self.agent_features(data_dict["inputs_m1"], n_total, len(lens), taps)
            return self._pyramid_and_heads(x, lens, pairwise, 0, taps)
        codes = self.encode_features(data_dict["inputs_m1"], n_total, taps)
        if taps is not None:
            taps["codes"] = codes
        hw = self.fh * self.fw
        return self.decode_features(codes, hw, n_total * hw, lens, pairwise, 0, taps)

    forward_with_encdec = forward
