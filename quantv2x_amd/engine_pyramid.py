"""Deployed W8A8 HEAL Pyramid-fusion path on MI355X: host side of ``libqv2x.so`` for
``heter_pyramid_collab_codebook_mc[_encdec]`` under ``QuantModel``.

What an agent runs before the link (``encode_features``, the reference's heter_pyramid_collab_codebook_mc_encdec.py:33-121):
    pillars -> PFN + scatter (a1, a2) -> ResNet BasicBlocks (QuantBasicBlock x 3, stride 2) -> codebook.encode (D = 64)
What the ego runs on the received indices (``decode_features``, :123-181):
    codebook.decode -> ResNeXt levels (QuantBottleneck x 3 / 5 / 8) on EVERY agent's map -> per level: occupancy head, score,
    weighted_fuse into the ego frame -> deblocks on the fused fp32 levels -> concat -> shrink_conv -> 1x1 heads

Activation CODES live in padded i8 BEV tensors, maps that are not on a quantizer grid (decoded feature, strided shortcut branch, fused
levels) are fp32 rows.  Every residual block is 3 launches (4 with a strided shortcut): conv1, conv2, [shortcut], conv3 with the
``+ shortcut -> ReLU -> block quantizer`` end fused into its epilogue.  There is no CPU / eager fallback."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from .engine import _ConvLayer, _DeconvLayer, _Heads, _dev, _pack_k4p, decode_tables, wave_section


def _q(state, name):
    return float(np.float32(state[name + "/a_delta"])), int(state[name + "/a_zp"])


class _Conv1x1:
    """One 1x1 QuantModule convolution on codes: the A fragments of v_mfma_i32_32x32x32_i8 + the epilogue constants."""

    def __init__(self, state, name, in_q, stride, dev):
        code = state[name + "/w_code"]
        cout, cin = code.shape[0], code.shape[1]
        if code.shape[2:] != (1, 1) or cout % 64 or cin % 32:
            raise NotImplementedError(f"{name}: a 1x1 convolution with cout % 64 == 0 and cin % 32 == 0")
        ws = code.reshape(cout, cin).astype(np.int64) - 128
        aw = 128 - state[name + "/w_zp"].astype(np.int64)
        ax = 128 - int(in_q[1])
        frag = ws.astype(np.int8).reshape(cout // 32, 32, cin // 32, 2, 16).transpose(0, 2, 3, 1, 4)   # [co/32][ci/32][half][co%32][16]
        self.w = _dev(frag, dev)
        self.scale = _dev((np.float32(in_q[0]) * state[name + "/w_delta"].astype(np.float32)).astype(np.float32), dev)
        corr = ax * ws.sum(axis=1) + cin * ax * aw
        assert np.abs(corr).max() < 2 ** 31
        self.corr, self.aw = _dev(corr.astype(np.int32), dev), _dev(aw.astype(np.int32), dev)
        self.bias = _dev(state[name + "/bias"].astype(np.float32), dev)
        self.name, self.cin, self.cout, self.stride = name, cin, cout, stride
        self.a_off = bool(state[name + "/a_off"])
        self.out_q = None if self.a_off else _q(state, name)


class _GConv:
    """The grouped 3x3 convolution of a bottleneck as block-diagonal 32 x 32 MFMA A fragments: [c/32][9 taps][lane][16 B]."""

    def __init__(self, state, name, in_q, stride, dev):
        code = state[name + "/w_code"]                                   # [c, cg, 3, 3]
        c, cg = code.shape[0], code.shape[1]
        if cg not in (4, 8, 16) or c % 32:
            raise NotImplementedError(f"{name}: grouped convolution with 4 / 8 / 16 channels per group, channels % 32 == 0")
        ws = code.astype(np.int64) - 128
        aw = 128 - state[name + "/w_zp"].astype(np.int64)
        ax = 128 - int(in_q[1])
        dense = np.zeros((c // 32, 9, 32, 32), np.int8)                 # [slab][tap][co][ci], zero outside the group
        taps = ws.reshape(c, cg, 9)
        for co in range(c):
            g0 = (co % 32) // cg * cg
            dense[co // 32, :, co % 32, g0:g0 + cg] = taps[co].T
        frag = dense.reshape(c // 32, 9, 32, 2, 16).transpose(0, 1, 3, 2, 4)      # [slab][tap][half][co][16]: lane = 32 * half + co
        self.w = _dev(frag, dev)
        self.scale = _dev((np.float32(in_q[0]) * state[name + "/w_delta"].astype(np.float32)).astype(np.float32), dev)
        corr = ax * ws.reshape(c, -1).sum(axis=1) + 9 * cg * ax * aw
        self.corr, self.aw = _dev(corr.astype(np.int32), dev), _dev(aw.astype(np.int32), dev)
        self.bias = _dev(state[name + "/bias"].astype(np.float32), dev)
        self.name, self.c, self.cg, self.stride = name, c, cg, stride
        self.out_q = _q(state, name)


class _DenseF32In(_DeconvLayer):
    """A QuantModule on an fp32 map: a deblock (ConvTranspose2d, k = s), or -- ``conv=True`` -- a 1x1 Conv2d whose [Cout, Cin, 1, 1]
    weight (scales per Cout) is laid out as the [Cin, Cout, 1, 1] deconvolution the kernel takes."""

    def __init__(self, state, name, dev, conv=False):
        if not conv:
            super().__init__(state, name, (1.0, 0), dev)
            return
        code = state[name + "/w_code"].astype(np.float32)
        dw = state[name + "/w_delta"].astype(np.float32).reshape(-1, 1, 1, 1)
        zw = state[name + "/w_zp"].astype(np.float32).reshape(-1, 1, 1, 1)
        wdeq = ((code - zw) * dw).astype(np.float32)                     # [Cout, Cin, 1, 1]
        cout, cin = wdeq.shape[:2]
        self.w = _dev(_pack_k4p(wdeq.reshape(cout, cin)), dev)           # rows = columns of the GEMM = cout
        self.bias = _dev(state[name + "/bias"].astype(np.float32), dev)
        self.name, self.cin, self.cout, self.s, self.in_q = name, cin, cout, 1, (1.0, 0)
        self.out_q = _q(state, name)


class _Occ:
    def __init__(self, state, name, in_q, dev):
        code = state[name + "/w_code"]
        if code.shape[0] != 1 or code.shape[2:] != (1, 1):
            raise NotImplementedError(f"{name}: a 1x1 convolution to ONE occupancy channel")
        ws = code.reshape(-1).astype(np.int64) - 128
        self.aw = int(128 - int(state[name + "/w_zp"][0]))
        ax = 128 - int(in_q[1])
        self.corr = int(ax * ws.sum() + ws.size * ax * self.aw)
        self.w = _dev(ws.astype(np.int8), dev)
        self.scale = float(np.float32(in_q[0]) * np.float32(state[name + "/w_delta"][0]))
        self.bias = float(np.float32(state[name + "/bias"][0]))
        self.name, self.c, self.out_q = name, ws.size, _q(state, name)
        da, za = np.float32(self.out_q[0]), np.float32(self.out_q[1])
        self.occ_values = ((np.arange(256, dtype=np.float32) - za) * da).astype(np.float32)          # fp32 occupancy of every code
        sig = (1.0 / (1.0 + np.exp(-self.occ_values.astype(np.float64)))).astype(np.float32)
        self.lut = _dev((sig + np.float32(1e-4)).astype(np.float32), dev)                               # sigmoid(occ) + 1e-4
        self.occ_lut = _dev(self.occ_values, dev)


class _Block:
    """One residual block: its convolutions, optional strided shortcut, and the quantizer after the add."""

    def __init__(self, state, name, in_q, stride, bottleneck, dev, f32_input=False):
        self.name, self.stride, self.bottleneck, self.f32_input = name, stride, bottleneck, f32_input
        if bottleneck:
            self.conv1 = _DenseF32In(state, name + ".conv1", dev, conv=True) if f32_input else _Conv1x1(state, name + ".conv1", in_q, 1, dev)
            self.conv2 = _GConv(state, name + ".conv2", self.conv1.out_q, stride, dev)
            self.conv3 = _Conv1x1(state, name + ".conv3", self.conv2.out_q, 1, dev)
            self.cout = self.conv3.cout
        else:
            self.conv1 = _ConvLayer(state, name + ".conv1", [(0, 64, in_q[0], in_q[1])], stride, dev)
            self.conv2 = _ConvLayer(state, name + ".conv2", [(0, self.conv1.cout, *self.conv1.out_q)], 1, dev)
            self.cout = self.conv2.cout
        self.down = None
        if (name + ".downsample/w_code") in state:
            if f32_input:
                raise NotImplementedError(f"{name}: a 1x1 shortcut on an fp32 input")
            self.down = _Conv1x1(state, name + ".downsample", in_q, stride, dev)
        self.in_q, self.out_q = in_q, _q(state, name)


class DeployedPyramidModel(nn.Module):
    """Same call contract as the reference's model: ``out = model(data_dict)`` (the hard ``encode -> decode`` path of
    ``forward_with_encdec``), plus ``encode_features`` / ``decode_features`` as the reference's encdec class exposes them."""

    def __init__(self, state: Dict[str, np.ndarray], device="cuda"):
        super().__init__()
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.Qv2xError("DeployedPyramidModel needs an MI355X (torch.cuda.is_available() is False)")
        if str(state.get("meta/fusion_method", "att")) != "pyramid":
            raise NotImplementedError("DeployedPyramidModel: a PTQ state exported from the Pyramid model")
        self.state, self.dev = state, torch.device(device)
        s, dev = state, self.dev
        self.nx, self.ny, _ = (int(v) for v in s["meta/grid"])
        self.hm, self.wm = (float(v) for v in s["meta/HW_metres"])
        self.ratio = float(s["meta/discrete_ratio"])
        self.has_codebook = bool(s["meta/has_codebook"])            # heter_pyramid_collab_mc (LiDAROnly/lidar_pyramid.yaml): no codebook

        n = "encoder_m1.pillar_vfe.pfn_layers.0.linear"
        wq = ((s[n + "/w_code"].astype(np.float32) - s[n + "/w_zp"].astype(np.float32)[:, None]) * s[n + "/w_delta"].astype(np.float32)[:, None]).astype(np.float32)
        if wq.shape != (64, 10):
            raise NotImplementedError("deployed PFN expects Linear(10 -> 64)")
        p = L.PfnParams()
        p.w[:] = wq.reshape(-1).tolist()
        p.b[:] = s[n + "/bias"].astype(np.float32).tolist()
        p.d1, p.z1 = float(np.float32(s[n + "/a_delta"])), float(s[n + "/a_zp"])
        p.d2, p.z2 = float(np.float32(s["pfn/a2_delta"])), float(s["pfn/a2_zp"])
        p.vox[:] = [float(np.float32(v)) for v in s["meta/voxel"]]
        p.off[:] = [float(np.float32(v)) for v in s["meta/offset"]]
        self.pfn = p
        q = (p.d2, int(p.z2))

        # ---- the agent's ResNet level ------------------------------------------------------------------------------------------
        nums, strides = [int(v) for v in s["meta/layer_nums"]], [int(v) for v in s["meta/layer_strides"]]
        self.agent_stride = strides[0]
        self.agent_blocks: List[_Block] = []
        for b in range(nums[0]):
            blk = _Block(s, f"backbone_m1.resnet.layer0.{b}", q, strides[0] if b == 0 else 1, False, dev)
            self.agent_blocks.append(blk)
            q = blk.out_q
        self.agent_q = q
        # ---- codebook (D = 64: its own encode kernel; other widths run the 256-wide one on zero-padded heads, see _level_blob) --
        self.native64 = True
        if self.has_codebook:
            # (seg_num > 1: the extended codebook [m * kc][D]; ``levels`` counts code planes = residual levels * m -- engine.py)
            self.enc_levels, self.segs = int(s["meta/codebook_levels"]), int(s.get("meta/codebook_segs", 1))
            self.levels = self.enc_levels * self.segs
            self.ke, self.D = (int(v) for v in s["codebook/0/codebook"].shape)
            self.kc = self.ke // self.segs
            if self.D > 256 or self.D % 4 or self.agent_blocks[-1].cout != self.D:
                raise NotImplementedError("deployed Pyramid codebook: width <= 256 equal to the agent feature's channels")
            self.native64 = self.D == 64
            if self.segs > 1 and self.D not in (64, 256):
                raise NotImplementedError("deployed Pyramid codebook: seg_num > 1 with a 64- or 256-wide codebook (other widths are zero-padded "
                                          "to 256, which moves the segment boundaries)")
            lut, lut_bias = decode_tables(s, self.enc_levels, self.D)
            lut = lut.reshape(self.levels, self.kc, self.D)
            self.lut, self.lut_bias = _dev(lut, dev), _dev(lut_bias, dev)
            self.level_blobs = [self._level_blob(l) for l in range(self.enc_levels)]
            self.level_ptrs = (C.c_void_p * self.enc_levels)(*[b.data_ptr() for b in self.level_blobs])
        # ---- the pyramid: ResNeXt levels, occupancy heads, deblocks -------------------------------------------------------------------
        p_nums, p_strides = [int(v) for v in s["meta/pyramid_layer_nums"]], [int(v) for v in s["meta/pyramid_layer_strides"]]
        self.ups = [int(v) for v in s["meta/upsample_strides"]]
        self.p_strides = p_strides
        self.pyr_blocks: List[List[_Block]] = []
        self.occ: List[_Occ] = []
        self.deblocks: List[_DenseF32In] = []
        q, cat_groups, c0 = (None if self.has_codebook else self.agent_q), [], 0
        for lvl in range(len(p_nums)):
            blocks = []
            for b in range(p_nums[lvl]):
                blk = _Block(s, f"pyramid_backbone.resnet.layer{lvl}.{b}", q, p_strides[lvl] if b == 0 else 1, True, dev, f32_input=q is None)
                blocks.append(blk)
                q = blk.out_q
            self.pyr_blocks.append(blocks)
            self.occ.append(_Occ(s, f"pyramid_backbone.single_head_{lvl}", q, dev))
            de = _DenseF32In(s, f"pyramid_backbone.deblocks.{lvl}.0", dev)
            self.deblocks.append(de)
            cat_groups.append((c0, de.cout, de.out_q[0], de.out_q[1]))
            c0 += de.cout
        self.cat_channels = c0
        self.shrink0 = _ConvLayer(s, "shrink_conv.layers.0.double_conv.0", cat_groups, 1, dev)
        self.shrink1 = _ConvLayer(s, "shrink_conv.layers.0.double_conv.1", [(0, self.shrink0.cout, *self.shrink0.out_q)], 1, dev)
        if self.shrink1.cout != 256:
            raise NotImplementedError("deployed path expects a 256-channel map in front of the heads")
        self.heads = _Heads(s, "", dev)
        self._bufs: Dict[tuple, dict] = {}
        # a level's occupancy head, fusion and deblock only feed the concat: they run on a side stream while the next level's blocks
        # run on the caller's (the ego side is ~70 launches on small maps; inside a captured HIP graph this is a fork / join)
        self.overlap_levels = True
        self._side = None
        self.fuse_blocks = True              # stride-1 bottlenecks with an identity shortcut as ONE launch (csrc/bottleneck_i8.hip)
        self.fuse_planes = (64, 128)

    # ------------------------------------------------------------------------------------------------------------------------------
    def _level_blob(self, l: int) -> torch.Tensor:
        """One level's heads in the layout of the encode kernel: the native 64-wide one (qv2x_codebook_encode64_f32), or -- other widths
        -- zero-padded to the 256-wide kernel's (every extra term of every fma chain is 0 * 0 and the padded |.|^2 chains are exactly 0,
        so the codes equal a native evaluation bit for bit)."""
        s, kc, D = self.state, self.ke, self.D                             # (kc: rows of the extended codebook)
        g = lambda n: s[f"codebook/{l}/{n}"].astype(np.float32)
        W = 64 if self.native64 else 256

        def pw(w):
            out = np.zeros((W, W), np.float32); out[:D, :D] = w; return _pack_k4p(out)

        def pb(b):
            out = np.zeros(W, np.float32); out[:D] = b; return out
        last = f"codebook/{l}/lhead_w" not in s
        cb = np.zeros((kc, W), np.float32); cb[:, :D] = g("codebook")
        parts = [pw(g("stage_w")), pb(g("stage_b")), pw(g("qhead_w")), pb(g("qhead_b")),
                 np.zeros((W // 4, W, 4), np.float32) if last else pw(g("lhead_w")), np.zeros(W, np.float32) if last else pb(g("lhead_b")),
                 _pack_k4p(cb), cb, np.zeros(kc, np.float32)]
        flat = np.concatenate([p.reshape(-1) for p in parts])
        wg = flat.size
        if not self.native64:                                              # the 256-wide kernel's blob carries the wave form's section too

            def full(w):
                out = np.zeros((W, W), np.float32); out[:D, :D] = w; return out
            flat = np.concatenate([flat, wave_section(full(g("stage_w")), full(g("qhead_w")), np.zeros((W, W), np.float32) if last else full(g("lhead_w")), cb, self.segs)])
        floats, c2fn = ((self.lib.qv2x_codebook64_level_floats, self.lib.qv2x_codebook64_c2_f32) if self.native64
                        else (self.lib.qv2x_codebook_level_floats, self.lib.qv2x_codebook_c2_f32))
        assert flat.size == floats(kc)
        blob = _dev(flat, self.dev)
        cb_off = wg - kc - kc * W
        L.check(c2fn(C.c_void_p(blob.data_ptr() + 4 * cb_off), kc, C.c_void_p(blob.data_ptr() + 4 * (wg - kc)), L.current_stream()), "codebook c2")
        return blob

    def _padded(self, n, h, w, c, zp):
        t = torch.empty((n, h + 2, w + 2, c), dtype=torch.int8, device=self.dev)
        t.fill_(int(zp) - 128)
        return t

    @staticmethod
    def _same_zp(qs, what):
        zps = {int(q[1]) for q in qs}
        if len(zps) != 1:
            raise NotImplementedError(f"{what}: the layers sharing this buffer must share a zero point (post-ReLU quantizers: 0)")
        return zps.pop()

    def _agent_ws(self, n: int) -> dict:
        key = ("agent", n)
        if key in self._bufs:
            return self._bufs[key]
        b = {}
        h, w, st = self.ny, self.nx, self.agent_stride
        self.fh, self.fw = (h - 1) // st + 1, (w - 1) // st + 1
        fh, fw = self.fh, self.fw
        b["canvas"] = self._padded(n, h, w, 64, self.pfn.z2)
        b["c1"] = self._padded(n, fh, fw, 64, self._same_zp([blk.conv1.out_q for blk in self.agent_blocks], "agent conv1"))
        zp = self._same_zp([blk.out_q for blk in self.agent_blocks], "agent blocks")
        b["x"] = [self._padded(n, fh, fw, 64, zp) for _ in range(2)]
        # the last block writes channels [0, D); with the 256-wide encode kernel the rest stay at the code of 0.0
        b["enc_in"] = self._padded(n, fh, fw, 64 if self.native64 else 256, zp)
        b["ds"] = torch.empty((n * fh * fw, 64), dtype=torch.float32, device=self.dev)
        if self.has_codebook:
            b["codes"] = torch.empty((self.levels, n, fh * fw), dtype=torch.uint8, device=self.dev)
        self._bufs[key] = b
        return b

    def _ego_ws(self, n: int, nb: int) -> dict:
        key = ("ego", n, nb)
        if key in self._bufs:
            return self._bufs[key]
        self._agent_ws(1)
        b = {"lvl": []}
        if self.has_codebook:
            b["feats"] = torch.empty((n * self.fh * self.fw, self.D), dtype=torch.float32, device=self.dev)
        h, w = self.fh, self.fw
        for lvl, blocks in enumerate(self.pyr_blocks):
            hi, wi = h, w
            st = self.p_strides[lvl]
            h, w = (h - 1) // st + 1, (w - 1) // st + 1
            width, planes = blocks[0].conv2.c, blocks[0].cout
            lv = {"h": h, "w": w, "hi": hi, "wi": wi}
            z1 = self._same_zp([blk.conv1.out_q for blk in blocks], f"level {lvl} conv1")
            lv["t1_in"] = self._padded(n, hi, wi, width, z1)                # conv1 of the first block runs at the input resolution
            lv["t1"] = self._padded(n, h, w, width, z1)
            lv["t2"] = self._padded(n, h, w, width, self._same_zp([blk.conv2.out_q for blk in blocks], f"level {lvl} conv2"))
            zx = self._same_zp([blk.out_q for blk in blocks], f"level {lvl} blocks")
            lv["x"] = [self._padded(n, h, w, planes, zx) for _ in range(2)]
            lv["ds"] = torch.empty((n * h * w, planes), dtype=torch.float32, device=self.dev)
            lv["score"] = torch.empty((n, h * w), dtype=torch.float32, device=self.dev)
            lv["occ_code"] = torch.empty((n, h * w), dtype=torch.uint8, device=self.dev)
            lv["fused"] = torch.empty((nb, h * w, planes), dtype=torch.float32, device=self.dev)
            b["lvl"].append(lv)
        fh, fw = b["lvl"][0]["h"] * self.ups[0], b["lvl"][0]["w"] * self.ups[0]
        for lvl, lv in enumerate(b["lvl"]):
            if (lv["h"] * self.ups[lvl], lv["w"] * self.ups[lvl]) != (fh, fw):
                raise ValueError(f"deblock {lvl} gives {lv['h'] * self.ups[lvl]}x{lv['w'] * self.ups[lvl]}, level 0 gives {fh}x{fw}: the grid does not "
                                 "line up across pyramid levels (the reference's torch.cat raises here)")
        self.oh, self.ow = fh, fw
        cat = torch.empty((nb, fh + 2, fw + 2, self.cat_channels), dtype=torch.int8, device=self.dev)
        c0 = 0
        for de in self.deblocks:
            cat[..., c0:c0 + de.cout] = int(de.out_q[1]) - 128
            c0 += de.cout
        b["cat"] = cat
        b["s0"] = self._padded(nb, fh, fw, self.shrink0.cout, self.shrink0.out_q[1])
        b["s1"] = self._padded(nb, fh, fw, self.shrink1.cout, self.shrink1.out_q[1])
        b["rows"] = torch.empty((nb * fh * fw, 256), dtype=torch.float32, device=self.dev)
        self._bufs[key] = b
        return b

    def work_inventory(self, n_agents: int, n_scenes: int) -> dict:
        """Algorithmic work of one forward of ``n_scenes`` scenes with ``n_agents`` agents in all, stage by stage (bench.py's roofline entries of the
        Pyramid model): multiply-accumulates of every layer = output pixels x the weight tensor's elements (a grouped 3x3: c x cg x 9 per
        pixel), and the bytes the memory-bound stages move."""
        s = self.state
        wsize = lambda name: int(np.prod(s[name + "/w_code"].shape))
        self._agent_ws(1)
        fh, fw = self.fh, self.fw
        agent = 0
        for blk in self.agent_blocks:                                   # conv1 (strided on the first block) and conv2 at the agent map's resolution
            agent += n_agents * fh * fw * (wsize(blk.name + ".conv1") + wsize(blk.name + ".conv2"))
            if blk.down is not None:
                agent += n_agents * fh * fw * wsize(blk.name + ".downsample")
        levels, h, w = [], fh, fw
        for lvl, blocks in enumerate(self.pyr_blocks):
            st = self.p_strides[lvl]
            ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
            macs = 0
            for i, blk in enumerate(blocks):
                hi, wi = (h, w) if i == 0 else (ho, wo)
                macs += n_agents * hi * wi * wsize(blk.name + ".conv1") + n_agents * ho * wo * (wsize(blk.name + ".conv2") + wsize(blk.name + ".conv3"))
                if blk.down is not None:
                    macs += n_agents * ho * wo * wsize(blk.name + ".downsample")
            planes = blocks[-1].cout
            levels.append({"int8_macs": macs, "blocks": len(blocks), "h": ho, "w": wo, "planes": planes,
                           # occupancy head + weighted_fuse: every agent's level map read (codes), the fused fp32 map written
                           "fuse_bytes": n_agents * ho * wo * planes + n_scenes * ho * wo * planes * 4,
                           "deblock_f32_macs": n_scenes * ho * wo * wsize(f"pyramid_backbone.deblocks.{lvl}.0")})
            h, w = ho, wo
        oh, ow = levels[0]["h"] * self.ups[0], levels[0]["w"] * self.ups[0]
        return {"agent_int8_macs": agent, "levels": levels,
                "encode_f32_macs": (n_agents * fh * fw * self.enc_levels * (3 * self.D * self.D + self.D * self.ke)) if self.has_codebook else 0,
                "decode_bytes": (n_agents * fh * fw * (self.levels + self.D * 4)) if self.has_codebook else 0,
                "shrink_int8_macs": n_scenes * oh * ow * (wsize(self.shrink0.name) + wsize(self.shrink1.name)),
                "heads_f32_macs": n_scenes * oh * ow * 256 * self.heads.cout}

    # ---- launch helpers ------------------------------------------------------------------------------------------------------------
    def _conv3x3(self, layer: _ConvLayer, x, n, h, w, out, out_q, res_mode=0, res=None, res_q=(0.0, 128)):
        d = L.ConvDesc()
        d.n, d.h, d.w, d.cin_total, d.stride, d.cout = n, h, w, x.shape[-1], layer.stride, layer.cout
        d.ngroups = len(layer.groups)
        for i, (c0, c, zx) in enumerate(layer.groups):
            d.group_c0[i], d.group_c[i], d.group_zx[i] = c0, c, zx
        d.out_ctotal, d.out_c0, d.relu = out.shape[-1], 0, 1
        d.out_delta, d.out_zp = out_q[0], float(out_q[1])
        st = L.current_stream()
        if res_mode:
            L.check(self.lib.qv2x_conv3x3_i8_res(C.byref(d), L.ptr(x), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr), L.ptr(layer.aw),
                                                 L.ptr(layer.bias), res_mode, L.ptr(res), int(res_q[1]), float(res_q[0]), L.ptr(out), st), layer.name)
            return
        if self.lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)):
            if layer.w_wide is None:
                layer.w_wide = torch.empty_like(layer.w)
                L.check(self.lib.qv2x_conv3x3_i8_pack_wide(C.byref(d), L.ptr(layer.w), L.ptr(layer.w_wide), st), layer.name)
            L.check(self.lib.qv2x_conv3x3_i8_wide(C.byref(d), L.ptr(x), L.ptr(layer.w_wide), L.ptr(layer.scale), L.ptr(layer.corr), L.ptr(layer.aw),
                                                  L.ptr(layer.bias), L.ptr(out), st), layer.name)
            return
        L.check(self.lib.qv2x_conv3x3_i8(C.byref(d), L.ptr(x), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr), L.ptr(layer.aw),
                                         L.ptr(layer.bias), L.ptr(out), st), layer.name)

    def _conv1x1(self, layer: _Conv1x1, x, n, h, w, out, mode, out_q=None, res=None, res_q=(0.0, 128)):
        d = L.Conv1x1Desc()
        d.n, d.h, d.w, d.cin, d.cout, d.stride = n, h, w, layer.cin, layer.cout, layer.stride
        d.mode, d.relu = mode, 1
        if mode != 1:
            d.out_ctotal, d.out_c0 = out.shape[-1], 0
            d.out_delta, d.out_zp = out_q[0], float(out_q[1])
        d.res_zx, d.res_delta = int(res_q[1]), float(res_q[0])
        L.check(self.lib.qv2x_conv1x1_i8(C.byref(d), L.ptr(x), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr), L.ptr(layer.aw), L.ptr(layer.bias),
                                         L.ptr(res), L.ptr(out), L.current_stream()), layer.name)

    def _gconv(self, layer: _GConv, x, n, h, w, out):
        d = L.GconvDesc()
        d.n, d.h, d.w, d.c, d.cg, d.stride, d.relu = n, h, w, layer.c, layer.cg, layer.stride, 1
        d.out_delta, d.out_zp = layer.out_q[0], float(layer.out_q[1])
        L.check(self.lib.qv2x_gconv3x3_i8(C.byref(d), L.ptr(x), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr), L.ptr(layer.aw), L.ptr(layer.bias),
                                          L.ptr(out), L.current_stream()), layer.name)

    def _dense_f32in(self, de: _DenseF32In, x, n, h, w, out, out_c0=0):
        d = L.DeconvDesc()
        d.n, d.h, d.w, d.cin, d.cout, d.s = n, h, w, de.cin, de.cout, de.s
        d.in_zx, d.in_delta = 0, 1.0
        d.out_ctotal, d.out_c0, d.relu = out.shape[-1], out_c0, 1
        d.out_delta, d.out_zp = de.out_q[0], float(de.out_q[1])
        d.out_h, d.out_w = out.shape[1] - 2, out.shape[2] - 2
        L.check(self.lib.qv2x_deconv_f32in(C.byref(d), L.ptr(x), L.ptr(de.w), L.ptr(de.bias), L.ptr(out), L.current_stream()), de.name)

    # ---- what an agent runs before the link ---------------------------------------------------------------------------------------------
    def pillars_to_canvas(self, inputs: dict, n: int):
        b = self._agent_ws(n)
        st = L.current_stream()
        vf = inputs["voxel_features"].contiguous()
        co = inputs["voxel_coords"].to(torch.int32).contiguous()
        npnt = inputs["voxel_num_points"].to(torch.int32).contiguous()
        if vf.dtype != torch.float32 or vf.dim() != 3 or tuple(vf.shape[1:]) != (32, 4):
            raise ValueError("voxel_features must be float32 [M, 32, 4]")
        canvas = b["canvas"]
        L.check(self.lib.qv2x_fill_i8(L.ptr(canvas), canvas.numel(), int(self.pfn.z2) - 128, st), "qv2x_fill_i8")
        L.check(self.lib.qv2x_pfn_scatter_i8(L.ptr(vf), L.ptr(co), L.ptr(npnt), vf.shape[0], 32, C.byref(self.pfn), L.ptr(canvas), n, self.ny, self.nx, st),
                "qv2x_pfn_scatter_i8")
        return canvas

    def agent_backbone(self, n: int, taps: Optional[dict] = None):
        b = self._agent_ws(n)
        x, h, w, xq = b["canvas"], self.ny, self.nx, (self.pfn.d2, int(self.pfn.z2))
        for i, blk in enumerate(self.agent_blocks):
            last = i == len(self.agent_blocks) - 1
            out = b["enc_in"] if last else b["x"][i % 2]
            self._conv3x3(blk.conv1, x, n, h, w, b["c1"], blk.conv1.out_q)
            if blk.down is not None:
                self._conv1x1(blk.down, x, n, h, w, b["ds"], 1)
                self._conv3x3(blk.conv2, b["c1"], n, self.fh, self.fw, out, blk.out_q, 2, b["ds"])
            else:
                self._conv3x3(blk.conv2, b["c1"], n, self.fh, self.fw, out, blk.out_q, 3, x, xq)
            if taps is not None:
                taps[blk.name + ".conv1"], taps[blk.name] = b["c1"].clone(), out[..., :blk.cout].clone()
            x, h, w, xq = out, self.fh, self.fw, blk.out_q
        return x

    def encode_codes(self, n: int, out: Optional[torch.Tensor] = None):
        b = self._agent_ws(n)
        codes = b["codes"] if out is None else out
        if codes.dtype != torch.uint8 or not codes.is_contiguous() or codes.numel() != self.levels * n * self.fh * self.fw:
            raise ValueError("encode_codes: out must be a contiguous uint8 tensor [levels, n_agents, H*W]")
        d = L.EncodeDesc()
        d.n, d.h, d.w, d.levels, d.kc, d.segs = n, self.fh, self.fw, self.enc_levels, self.kc, self.segs
        d.in_zx, d.in_delta = int(self.agent_q[1]), float(self.agent_q[0])
        if self.native64:
            L.check(self.lib.qv2x_codebook_encode64_f32(C.byref(d), b["enc_in"].shape[-1], L.ptr(b["enc_in"]), self.level_ptrs, L.ptr(codes),
                                                        L.current_stream()), "qv2x_codebook_encode64_f32")
        else:
            L.check(self.lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(b["enc_in"]), self.level_ptrs, L.ptr(codes), L.current_stream()),
                    "qv2x_codebook_encode_f32")
        return codes

    @torch.no_grad()
    def encode_features(self, inputs: dict, n: int, taps: Optional[dict] = None, out: Optional[torch.Tensor] = None):
        """codes u8 [levels, n, H*W] of ``n`` agents (heter_pyramid_collab_codebook_mc_encdec.py:33-121)."""
        if not self.has_codebook:
            raise L.Qv2xError("encode_features: this model has no codebook (no wire format); call forward")
        canvas = self.pillars_to_canvas(inputs, n)
        if taps is not None:
            taps["canvas"] = canvas
        self.agent_backbone(n, taps)
        return self.encode_codes(n, out)

    # ---- what the ego runs on the received codes ------------------------------------------------------------------------------------------
    def _bottleneck(self, blk: _Block, x, xq, n, h, w, out):
        """a stride-1 bottleneck with an identity shortcut: conv1 -> grouped 3x3 -> conv3 + x, one launch (qv2x_bottleneck_i8)"""
        d = L.BottleneckDesc()
        d.n, d.h, d.w, d.planes, d.width, d.cg = n, h, w, blk.cout, blk.conv2.c, blk.conv2.cg
        d.in_zx, d.in_delta = int(xq[1]), float(xq[0])
        d.delta1, d.zp1 = blk.conv1.out_q[0], float(blk.conv1.out_q[1])
        d.delta2, d.zp2 = blk.conv2.out_q[0], float(blk.conv2.out_q[1])
        d.out_delta, d.out_zp = blk.out_q[0], float(blk.out_q[1])
        layers = (blk.conv1, blk.conv2, blk.conv3)
        arr = lambda name: (C.c_void_p * 3)(*[getattr(l, name).data_ptr() for l in layers])
        L.check(self.lib.qv2x_bottleneck_i8(C.byref(d), L.ptr(x), arr("w"), arr("scale"), arr("corr"), arr("aw"), arr("bias"), L.ptr(out),
                                            L.current_stream()), blk.name)

    def _block(self, blk: _Block, x, xq, n, h, w, lv, out, feats_f32=None):
        """One bottleneck: ``x`` codes at (h, w) (or the fp32 decoded map), result codes in ``out`` at the level's resolution."""
        ho, wo = lv["h"], lv["w"]
        # one launch where it pays: 31 vs 45 us (64 planes, 70 400 cells), 29 vs 32 us (128 planes); at 256 planes / 4 400 cells the fused
        # kernel has 78 workgroups of three serial phases and loses (41 vs 27 us), so that level stays on the per-layer kernels
        if self.fuse_blocks and blk.stride == 1 and blk.down is None and not blk.f32_input and blk.cout in self.fuse_planes and blk.conv2.c == 2 * blk.cout:
            return self._bottleneck(blk, x, xq, n, h, w, out)
        t1 = lv["t1_in"] if (h, w) != (ho, wo) else lv["t1"]
        if blk.f32_input:
            self._dense_f32in(blk.conv1, feats_f32, n, h, w, t1)
        else:
            self._conv1x1(blk.conv1, x, n, h, w, t1, 0, blk.conv1.out_q)
        self._gconv(blk.conv2, t1, n, h, w, lv["t2"])
        if blk.down is not None:
            self._conv1x1(blk.down, x, n, h, w, lv["ds"], 1)
            self._conv1x1(blk.conv3, lv["t2"], n, ho, wo, out, 2, blk.out_q, lv["ds"])
        elif blk.f32_input:
            self._conv1x1(blk.conv3, lv["t2"], n, ho, wo, out, 2, blk.out_q, feats_f32)
        else:
            self._conv1x1(blk.conv3, lv["t2"], n, ho, wo, out, 3, blk.out_q, x, xq)

    def _fuse_level(self, lvl, lv, x, xq, n, h, w, lens, pairwise, ego, cat, c0):
        """occupancy head + score, weighted_fuse of every scene, the level's deblock into the concat; returns the fp32 occupancy maps"""
        st = L.current_stream
        oc = self.occ[lvl]
        d = L.OccDesc()
        d.n, d.h, d.w, d.c, d.aw, d.corr = n, h, w, oc.c, oc.aw, oc.corr
        d.scale, d.bias, d.out_delta, d.out_zp = oc.scale, oc.bias, oc.out_q[0], float(oc.out_q[1])
        L.check(self.lib.qv2x_occ_score_i8(C.byref(d), L.ptr(x), L.ptr(oc.w), L.ptr(oc.lut), L.ptr(lv["score"]), L.ptr(lv["occ_code"]), st()), oc.name)
        occ = oc.occ_lut[lv["occ_code"].long()].view(n, 1, h, w)
        start = 0
        for bi, na in enumerate(lens):
            fd = self._fuse_desc(na, h, w, pairwise.shape[1], ego)
            L.check(self.lib.qv2x_pyramid_weighted_fuse_i8(C.byref(fd), self.pyr_blocks[lvl][-1].cout, L.ptr(x[start:start + na]), int(xq[1]), float(xq[0]),
                                                           L.ptr(lv["score"][start:start + na]), L.ptr(pairwise[bi]), L.ptr(lv["fused"][bi]), st()),
                    "qv2x_pyramid_weighted_fuse_i8")
            start += na
        self._dense_f32in(self.deblocks[lvl], lv["fused"], len(lens), h, w, cat, c0)
        return occ

    def _fuse_desc(self, n, h, w, max_cav, ego):
        d = L.FuseDesc()
        d.agents, d.h, d.w, d.levels, d.kc = n, h, w, 1, 1
        d.max_cav, d.ego = max_cav, ego
        d.h_metres, d.w_metres, d.discrete_ratio = self.hm, self.wm, self.ratio
        return d

    @torch.no_grad()
    def decode_features(self, codes, agent_stride: int, level_stride: int, lens: List[int], pairwise: torch.Tensor, ego: int = 0,
                        taps: Optional[dict] = None) -> dict:
        """``codes``: pointer / tensor of u8 planes, agent a's level-l plane at ``a * agent_stride + l * level_stride``; ``lens`` agents per
        scene; ``pairwise`` f64 [B, L, L, 4, 4] (heter_pyramid_collab_codebook_mc_encdec.py:123-181)."""
        if not self.has_codebook:
            raise L.Qv2xError("decode_features: this model has no codebook (no wire format); call forward")
        n, nb = sum(lens), len(lens)
        b = self._ego_ws(n, nb)
        hw = self.fh * self.fw
        cptr = codes if isinstance(codes, C.c_void_p) else L.ptr(codes)
        L.check(self.lib.qv2x_codebook_decode_f32(cptr, agent_stride, level_stride, n, hw, self.levels, self.kc, self.D, L.ptr(self.lut), L.ptr(self.lut_bias),
                                                  L.ptr(b["feats"]), L.current_stream()), "qv2x_codebook_decode_f32")
        return self._pyramid_and_heads(None, None, b["feats"], lens, pairwise, ego, taps)

    def _pyramid_and_heads(self, x, xq, feats_f32, lens: List[int], pairwise: torch.Tensor, ego: int, taps: Optional[dict]) -> dict:
        """The ResNeXt levels on every agent's map -- codes ``x`` with quantizer ``xq``, or the decoded fp32 map -- fusion, deblocks,
        shrink_conv and the heads."""
        n, nb = sum(lens), len(lens)
        b = self._ego_ws(n, nb)
        st = L.current_stream
        h, w = self.fh, self.fw
        occ_maps, c0 = [], 0
        for lvl, blocks in enumerate(self.pyr_blocks):
            lv = b["lvl"][lvl]
            for i, blk in enumerate(blocks):
                out = lv["x"][i % 2]
                self._block(blk, x, xq, n, h, w, lv, out, feats_f32)
                x, xq, h, w = out, blk.out_q, lv["h"], lv["w"]
                if taps is not None:
                    taps[blk.name] = out.clone()
            if self.overlap_levels:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                main = torch.cuda.current_stream()
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    occ_maps.append(self._fuse_level(lvl, lv, x, xq, n, h, w, lens, pairwise, ego, b["cat"], c0))
            else:
                occ_maps.append(self._fuse_level(lvl, lv, x, xq, n, h, w, lens, pairwise, ego, b["cat"], c0))
            c0 += self.deblocks[lvl].cout
            if taps is not None:
                torch.cuda.current_stream().wait_stream(self._side) if self.overlap_levels else None
                taps[f"occ_code{lvl}"], taps[f"score{lvl}"], taps[f"fused{lvl}"] = lv["occ_code"].clone(), lv["score"].clone(), lv["fused"].clone()
        if self.overlap_levels:
            torch.cuda.current_stream().wait_stream(self._side)
        oh, ow = self.oh, self.ow
        self._conv3x3(self.shrink0, b["cat"], nb, oh, ow, b["s0"], self.shrink0.out_q)
        self._conv3x3(self.shrink1, b["s0"], nb, oh, ow, b["s1"], self.shrink1.out_q)
        q = self.shrink1.out_q
        L.check(self.lib.qv2x_dequant_i8_f32(L.ptr(b["s1"]), nb, oh, ow, 256, int(q[1]), float(q[0]), L.ptr(b["rows"]), st()), "qv2x_dequant_i8_f32")
        hd = self.heads
        preds = torch.empty((nb, hd.cout, oh, ow), dtype=torch.float32, device=self.dev)
        L.check(self.lib.qv2x_heads_f32(L.ptr(b["rows"]), nb * oh * ow, oh * ow, hd.cout, hd.cout_pad, L.ptr(hd.w), L.ptr(hd.bias), L.ptr(hd.da), L.ptr(hd.za),
                                        L.ptr(preds), st()), "qv2x_heads_f32")
        if taps is not None:
            taps["cat"], taps[self.shrink0.name], taps[self.shrink1.name], taps["features"] = b["cat"], b["s0"], b["s1"], feats_f32
        c, r, _ = hd.splits
        return {"pyramid": "collab", "cls_preds": preds[:, :c], "reg_preds": preds[:, c:c + r], "dir_preds": preds[:, c + r:], "preds_tensor": preds,
                "occ_single_list": occ_maps}

    # ---- stage interface of the multi-GPU driver (quantv2x_amd/dist.py): one agent per rank, code planes + pose over the link ----------
    def wire_shape(self):
        self._agent_ws(1)
        return self.levels, self.fh * self.fw

    def encode_into(self, inputs: dict, frames: int, codes_out: torch.Tensor):
        """``encode_features`` for ``frames`` frames of ONE agent (batch index = frame) into ``codes_out`` u8 [levels, frames, H*W]."""
        return self.encode_features(inputs, frames, out=codes_out)

    def pairwise_from_poses(self, gathered: torch.Tensor, world: int, agent_stride: int, pose_offset: int, max_cav: int, out: torch.Tensor):
        L.check(self.lib.qv2x_pairwise_from_poses_f64(L.ptr(gathered), world, agent_stride, pose_offset, max_cav, L.ptr(out), L.current_stream()),
                "qv2x_pairwise_from_poses_f64")

    def fuse_frames_and_heads(self, gathered: torch.Tensor, agent_stride: int, level_stride: int, frame_stride: int, pairwise: torch.Tensor,
                              n_agents: int, ego: int, own_codes: Optional[torch.Tensor], frames: int) -> dict:
        """``decode_features`` for ``frames`` scenes whose agents' code planes lie ``agent_stride`` bytes apart in ``gathered`` (frame f at
        ``+ f * frame_stride``); ``pairwise`` f64 [frames, L, L, 4, 4]; this rank's agent is ``ego``."""
        outs = [self.decode_features(C.c_void_p(gathered.data_ptr() + f * frame_stride), agent_stride, level_stride, [n_agents],
                                     pairwise[f:f + 1], ego) for f in range(frames)]
        if frames == 1:
            return outs[0]
        preds = torch.cat([o["preds_tensor"] for o in outs])
        c, r, _ = self.heads.splits
        return {"pyramid": "collab", "cls_preds": preds[:, :c], "reg_preds": preds[:, c:c + r], "dir_preds": preds[:, c + r:], "preds_tensor": preds,
                "occ_single_list": [torch.cat([o["occ_single_list"][l] for o in outs]) for l in range(len(self.occ))]}

    # ---- the reference's model contract ------------------------------------------------------------------------------------------------
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
            if isinstance(rl, torch.Tensor) and rl.is_cuda and torch.cuda.is_current_stream_capturing():
                raise ValueError("record_len on the GPU cannot be read during HIP-graph capture: pass a CPU tensor")
            lens = [int(v) for v in (rl.tolist() if isinstance(rl, torch.Tensor) else rl)]
        if not self.has_codebook:                                    # heter_pyramid_collab_mc: the agents' codes go straight into the pyramid
            canvas = self.pillars_to_canvas(data_dict["inputs_m1"], n_total)
            if taps is not None:
                taps["canvas"] = canvas
            x = self.agent_backbone(n_total, taps)
            return self._pyramid_and_heads(x, self.agent_q, None, lens, pairwise, 0, taps)
        codes = self.encode_features(data_dict["inputs_m1"], n_total, taps)
        hw = self.fh * self.fw
        if taps is not None:
            taps["codes"] = codes
        return self.decode_features(codes, hw, n_total * hw, lens, pairwise, 0, taps)

    forward_with_encdec = forward

    def capture(self, data_dict: dict):
        """One frame as a HIP graph (see ``DeployedModel.capture``: fixed pillar count, CPU ``record_len``)."""
        rl = data_dict.get("record_len")
        if data_dict["pairwise_t_matrix"].shape[0] > 1 and isinstance(rl, torch.Tensor) and rl.is_cuda:
            data_dict = dict(data_dict, record_len=rl.cpu())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.forward(data_dict)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.forward(data_dict)

        def replay():
            graph.replay()
            return out
        replay.graph = graph
        return replay


class DeployedHeterPyramidModel(nn.Module):
    """HEAL's heterogeneous Pyramid scene (heter_pyramid_collab_codebook_mc.py:45-86, 164-249: one encoder / ResNet backbone / aligner per
    modality, the agents' 64-channel maps assembled in ``agent_modality_list`` order, shared codebook / PyramidFusion / shrink_conv / heads).
    One ``DeployedPyramidModel`` per modality (``export_ptq_state(qt, modality=m)``: its own agent-side weights and quantizers) runs
    ``encode_features`` on that modality's agents; the code planes go into one buffer in agent order -- the wire format IS the interface --
    and the ego modality's engine runs ``decode_features`` (round 5; the baseline model's twin is ``engine.DeployedHeterModel``)."""

    def __init__(self, states: Dict[str, Dict[str, np.ndarray]], ego_modality: Optional[str] = None, device="cuda"):
        super().__init__()
        if not states:
            raise ValueError("DeployedHeterPyramidModel: no modality")
        self.engines = {m: DeployedPyramidModel(st, device=device) for m, st in states.items()}
        self.modality_names = list(states)                               # the model's modality_name_list order (deploy passes it that way)
        self.main = self.engines[ego_modality if ego_modality in self.engines else next(iter(self.engines))]
        for m, e in self.engines.items():
            e._agent_ws(1)
            if not e.has_codebook:
                raise NotImplementedError("deployed heterogeneous Pyramid path: codebook models (the code planes are what the modalities share)")
            if (e.fh, e.fw, e.levels, e.kc, e.D) != (self.main.fh, self.main.fw, self.main.levels, self.main.kc, self.main.D):
                raise ValueError(f"modality {m}: feature map {e.fh} x {e.fw} / codebook {e.levels} x {e.kc} x {e.D} differs from the ego modality's")
        self.dev = self.main.dev
        self._slots: Dict[tuple, Dict[str, torch.Tensor]] = {}

    @torch.no_grad()
    def forward(self, data_dict: dict, taps: Optional[dict] = None) -> dict:
        # the reference's heter_pyramid_collab takes a TENSOR of 1-based modality codes and maps it onto modality_name_list
        # (heter_pyramid_collab.py:143-150; the plugin mirror's _named_modalities does the same); names pass through
        aml = data_dict["agent_modality_list"]
        aml = aml.tolist() if isinstance(aml, torch.Tensor) else list(aml)
        agents = [self.modality_names[int(a) - 1] if isinstance(a, (int, np.integer)) or (isinstance(a, float) and a == int(a)) else str(a) for a in aml]
        n_total, hw, lv = len(agents), self.main.fh * self.main.fw, self.main.levels
        unknown = sorted(set(agents) - set(self.engines))
        if unknown:
            raise NotImplementedError(f"deployed heterogeneous Pyramid path: no engine for modality {unknown}")
        slots = self._slots.get(tuple(agents))
        if slots is None:
            if torch.cuda.is_current_stream_capturing():
                raise L.Qv2xError(f"DeployedHeterPyramidModel: agent layout {agents} is new and the stream is capturing; run one eager forward first")
            slots = {m: torch.as_tensor([i for i, a in enumerate(agents) if a == m], dtype=torch.int64, device=self.dev) for m in self.engines}
            self._slots[tuple(agents)] = slots
        pairwise = data_dict["pairwise_t_matrix"]
        if pairwise.dtype != torch.float64 or not pairwise.is_contiguous():
            pairwise = pairwise.to(torch.float64).contiguous()
        if pairwise.shape[0] == 1:
            lens = [n_total]
        else:
            rl = data_dict["record_len"]
            lens = [int(v) for v in (rl.tolist() if isinstance(rl, torch.Tensor) else rl)]
        enc = torch.empty((lv, n_total * hw), dtype=torch.uint8, device=self.dev)
        for m, eng in self.engines.items():
            k = int(slots[m].numel())
            if not k:
                continue
            mt = {} if taps is not None else None
            codes = eng.encode_features(data_dict["inputs_" + m], k, mt).view(lv, k, hw)
            enc.view(lv, n_total, hw).index_copy_(1, slots[m], codes)          # agent order (a copy, no arithmetic)
            if taps is not None:
                taps["modality/" + m] = mt
        if taps is not None:
            taps["codes"] = enc.view(lv, n_total, hw)
        return self.main.decode_features(enc, hw, n_total * hw, lens, pairwise, 0, taps)

    forward_with_encdec = forward


def deploy_heter_pyramid(qt_model, device="cuda") -> DeployedHeterPyramidModel:
    from .ptq_state import export_ptq_state
    model = qt_model.model if hasattr(qt_model, "model") else qt_model
    states = {m: export_ptq_state(qt_model, modality=m) for m in model.modality_name_list}
    return DeployedHeterPyramidModel(states, ego_modality=getattr(model, "ego_modality", None), device=device)
