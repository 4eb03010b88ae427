"""TEST INFRASTRUCTURE: the deployed W8A8 hot path restated on the CPU, stage by stage.

Consumes the plain-numpy PTQ state (``quantv2x_amd.ptq_state.export_ptq_state`` output -- data only) and a
numpy scene (``quantv2x_amd.synth.make_scene``).  Integer / fixed-order arithmetic is in
``qv2x_oracle.c`` (built by ``oracle/Makefile``); geometry and attention are numpy (``geometry.py``).

Stage -> reference function:
  pfn_scatter  a1+a2   QuantPillarVFE/QuantPFNLayer (quant_block.py:589-715) + PointPillarScatter
  backbone     a3+a5   QuantBaseBEVBackbone (quant_block.py:243-303) over QuantModule (quant_layer.py:391-410)
  shrinker     a4+a5   QuantDownsampleConv (quant_block.py:552-586)
  encode       a6      UMGMQuantizer.encode (codebook.py:330-337)
  decode       a7      UMGMQuantizer.decode (codebook.py:339-343) as three table look-ups
  fuse         a8-a10  normalize_pairwise_tfm + warp_affine_simple + AttFusion
  heads        a11     1x1 QuantModule heads
"""
import ctypes
import os
import subprocess

import numpy as np

from . import geometry

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_f32p = ctypes.POINTER(ctypes.c_float)


def lib():
    global _LIB
    if _LIB is None:
        # QV2X_ORACLE_LIB: another build of the same checker (tests: the sanitizer build `make -C oracle asan`, the division-form build)
        path = os.environ.get("QV2X_ORACLE_LIB") or os.path.join(_HERE, "_build", "libqv2x_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, ct=None):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def _cf(x):
    return ctypes.c_float(float(x))


def decode_tables(state, levels=3):
    """decode(c0, c1, c2) = bias + T0[c0] + T1[c1] + T2[c2] (all heads are affine).  float64 algebra, fp32 result."""
    L = levels
    g = lambda l, n: state[f"codebook/{l}/{n}"].astype(np.float64)
    # level L-1 has no side head:  r_l = Wr_l (Wd_l C_l[c] + bd_l + Ws_l r_{l+1} + bs_l) + br_l
    carry_m = None   # matrix applied to deeper levels' result
    tables, const = [None] * L, np.zeros(256)
    chain = np.eye(256)
    for l in range(L):
        wr, br = g(l, "restore_w"), g(l, "restore_b")
        wd, bd = g(l, "dqhead_w"), g(l, "dqhead_b")
        front = chain @ wr
        tables[l] = (front @ wd @ g(l, "codebook").T).T                # [k, 256]
        const = const + front @ bd + chain @ br
        if l < L - 1:
            ws, bs = g(l, "side_w"), g(l, "side_b")
            const = const + front @ bs
            chain = front @ ws
    return np.stack(tables).astype(np.float32), const.astype(np.float32)


class Oracle:
    def __init__(self, state):
        self.s = state
        self.nx, self.ny, _ = (int(v) for v in state["meta/grid"])
        self.names = [str(n) for n in state["meta/module_names"]]
        self.layer_nums = [int(v) for v in state["meta/layer_nums"]]
        self.strides = [int(v) for v in state["meta/layer_strides"]]
        self.ups = [int(v) for v in state["meta/upsample_strides"]]
        self.has_codebook = bool(state["meta/has_codebook"])
        if self.has_codebook:
            self.lut, self.lut_bias = decode_tables(state, int(state["meta/codebook_levels"]))

    # ---- helpers -------------------------------------------------------------------------------
    def q(self, name):
        s = self.s
        return dict(code=s[name + "/w_code"], dw=s[name + "/w_delta"], zw=s[name + "/w_zp"], bias=s[name + "/bias"],
                    da=np.float32(s[name + "/a_delta"]), za=np.float32(s[name + "/a_zp"]), a_off=bool(s[name + "/a_off"]))

    @staticmethod
    def dequant_weight(p):
        shape = [-1] + [1] * (p["code"].ndim - 1)
        return ((p["code"].astype(np.float32) - p["zw"].reshape(shape)) * p["dw"].reshape(shape)).astype(np.float32)

    # ---- a1 + a2 -------------------------------------------------------------------------------
    def pfn_scatter(self, scene, n_agents):
        s = self.s
        p = self.q("encoder_m1.pillar_vfe.pfn_layers.0.linear")
        w = _f32(self.dequant_weight(p))
        inp = scene["inputs_m1"]
        vf, co, npt = _f32(inp["voxel_features"]), _i32(inp["voxel_coords"]), _i32(inp["voxel_num_points"])
        M, P = vf.shape[0], vf.shape[1]
        codes = np.zeros((M, 64), np.uint8)
        vox, off = _f32(s["meta/voxel"]), _f32(s["meta/offset"])
        d2, z2 = np.float32(s["pfn/a2_delta"]), np.float32(s["pfn/a2_zp"])
        lib().orc_pfn(_p(vf), _p(co), _p(npt), M, P, _p(w), _p(_f32(p["bias"])), _cf(p["da"]), _cf(p["za"]),
                      _cf(d2), _cf(z2), _p(vox), _p(off), _p(codes))
        canvas = np.full((n_agents, self.ny, self.nx, 64), int(z2), np.uint8)
        lib().orc_scatter(_p(codes), _p(co), M, self.ny, self.nx, _p(canvas))
        return codes, canvas, (d2, int(z2))

    # ---- a3 / a4 -------------------------------------------------------------------------------
    def conv(self, name, x, in_q, stride=1, out=None, out_c0=0):
        """x u8 [N,H,W,Cin]; in_q = list of (c0, c, delta, zp) channel groups.  Returns (codes, (delta, zp))."""
        p = self.q(name)
        n, h, w, cin = x.shape
        cout = p["code"].shape[0]
        g = len(in_q)
        scale = np.stack([np.float32(d) * p["dw"].astype(np.float32) for (_, _, d, _) in in_q]).astype(np.float32)
        ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        if out is None:
            out = np.zeros((n, ho, wo, cout), np.uint8)
        lib().orc_conv3x3(_p(np.ascontiguousarray(x)), n, h, w, cin, stride, g,
                          _p(_i32([q[0] for q in in_q])), _p(_i32([q[1] for q in in_q])), _p(_i32([q[3] for q in in_q])),
                          _p(np.ascontiguousarray(p["code"])), _p(_i32(p["zw"])), cout, _p(scale), _p(_f32(p["bias"])),
                          1, _cf(p["da"]), _cf(p["za"]), _p(out), out.shape[3], out_c0)
        return out, (p["da"], int(p["za"]))

    def deconv(self, name, x, in_q, s, out, out_c0):
        p = self.q(name)
        n, h, w, cin = x.shape
        wdeq = _f32(self.dequant_weight(p))                  # [Cin, Cout, s, s], per-C_in scales
        cout = wdeq.shape[1]
        lib().orc_deconv(_p(np.ascontiguousarray(x)), n, h, w, cin, _cf(in_q[0]), int(in_q[1]), _p(wdeq), _p(_f32(p["bias"])),
                         cout, s, 1, _cf(p["da"]), _cf(p["za"]), _p(out), out.shape[3], out_c0)
        return (p["da"], int(p["za"]))

    def backbone(self, canvas, canvas_q, taps=None):
        x, xq = canvas, canvas_q
        n = x.shape[0]
        cat, cat_q, c0 = None, [], 0
        for lvl in range(len(self.layer_nums)):
            for i in range(self.layer_nums[lvl] + 1):
                name = f"backbone_m1.blocks.{lvl}.{i + 1}"
                x, xq = self.conv(name, x, [(0, x.shape[3], xq[0], xq[1])], stride=self.strides[lvl] if i == 0 else 1)
                if taps is not None:
                    taps[name] = x
            s = self.ups[lvl]
            name = f"backbone_m1.deblocks.{lvl}.0"
            cup = self.s[name + "/w_code"].shape[1]
            if cat is None:
                total = sum(self.s[f"backbone_m1.deblocks.{l}.0/w_code"].shape[1] for l in range(len(self.ups)))
                cat = np.zeros((n, x.shape[1] * s, x.shape[2] * s, total), np.uint8)
            oq = self.deconv(name, x, xq, s, cat, c0)
            cat_q.append((c0, cup, oq[0], oq[1]))
            if taps is not None:
                taps[name] = cat[..., c0:c0 + cup]
            c0 += cup
        return cat, cat_q

    def shrinker(self, cat, cat_q, taps=None):
        x, xq = self.conv("shrinker_m1.layers.0.double_conv.0", cat, cat_q)
        if taps is not None:
            taps["shrinker_m1.layers.0.double_conv.0"] = x
        y, yq = self.conv("shrinker_m1.layers.0.double_conv.1", x, [(0, x.shape[3], xq[0], xq[1])])
        if taps is not None:
            taps["shrinker_m1.layers.0.double_conv.1"] = y
        return y, yq

    # ---- a6 / a7 -------------------------------------------------------------------------------
    def encode_rows(self, rows, want_gaps=False):
        L = int(self.s["meta/codebook_levels"])
        R = rows.shape[0]
        keep = []

        def arr(tag):
            ptrs = (ctypes.c_void_p * L)()
            for l in range(L):
                key = f"codebook/{l}/{tag}"
                if key in self.s:
                    a = _f32(self.s[key]); keep.append(a); ptrs[l] = a.ctypes.data
                else:
                    ptrs[l] = None
            return ptrs
        # seg_num (m) segments: the state holds the EXTENDED codebook [m * kc][256] (ptq_state.extended_codebook); planes = L * m
        S = int(self.s.get("meta/codebook_segs", 1))
        kc = self.s["codebook/0/codebook"].shape[0] // S
        codes = np.zeros((L * S, R), np.uint8)
        gaps = np.zeros((L * S, R), np.float32) if want_gaps else None
        lib().orc_codebook_encode_seg(_p(_f32(rows)), R, L, kc, 256, S, arr("stage_w"), arr("stage_b"), arr("qhead_w"), arr("qhead_b"),
                                      arr("lhead_w"), arr("lhead_b"), arr("codebook"), _p(codes),
                                      _p(gaps) if want_gaps else None)
        return (codes, gaps) if want_gaps else codes

    def encode(self, feat, feat_q):
        rows = ((feat.astype(np.float32) - np.float32(feat_q[1])) * np.float32(feat_q[0])).reshape(-1, feat.shape[-1])
        return self.encode_rows(rows)

    def decode(self, codes):
        """codes u8 [planes = L * m][R]: plane (l, s) looks up row s * kc + code of level l's table = row code of plane l * m + s."""
        P, R = codes.shape
        out = np.zeros((R, 256), np.float32)
        kc = self.lut.shape[0] * self.lut.shape[1] // P
        lib().orc_decode_lut(_p(np.ascontiguousarray(codes)), R, P, kc, _p(_f32(self.lut)), _p(_f32(self.lut_bias)), _p(out))
        return out

    # ---- a8 - a11 ------------------------------------------------------------------------------
    def fuse(self, feats, pairwise_t, record_len):
        """feats f32 [sum_N, h, w, 256]; returns [B, h, w, 256]."""
        H, W = (float(v) for v in self.s["meta/HW_metres"])
        affine = geometry.normalize_pairwise_tfm(np.asarray(pairwise_t), H, W, float(self.s["meta/discrete_ratio"]))
        out, start = [], 0
        for b, n in enumerate(int(v) for v in record_len):
            warped = geometry.warp_to_ego(feats[start:start + n], affine[b], n)
            out.append(geometry.max_fuse(warped) if str(self.s.get("meta/fusion_method", "att")) == "max" else geometry.att_fuse(warped))
            start += n
        return np.stack(out)

    def heads(self, fused, suffix=""):
        """fused [B, h, w, 256] -> preds [B, C, h, w] in cls, reg, dir order."""
        b, h, w, c = fused.shape
        rows = _f32(fused.reshape(-1, c))
        outs = []
        for head in ("cls_head", "reg_head", "dir_head"):
            p = self.q(head + suffix)
            wdeq = _f32(self.dequant_weight(p).reshape(p["code"].shape[0], -1))
            o = np.zeros((rows.shape[0], wdeq.shape[0]), np.float32)
            lib().orc_heads(_p(rows), rows.shape[0], c, _p(wdeq), _p(_f32(p["bias"])), wdeq.shape[0],
                            0 if p["a_off"] else 1, _cf(p["da"]), _cf(p["za"]), _p(o))
            outs.append(o.reshape(b, h, w, -1).transpose(0, 3, 1, 2))
        return outs

    # ---- whole path ----------------------------------------------------------------------------
    def canvas(self, scene, n_agents, taps=None):
        """a1 + a2 (PointPillar) or a13 (SECOND: oracle/spec_second.py) on ``scene["inputs_m1"]``: the uint8 BEV canvas and its quantizer."""
        taps = {} if taps is None else taps
        if "meta/encoder" in self.s and str(self.s["meta/encoder"]) == "second":
            from .spec_second import OracleSecond
            sec = OracleSecond(self.s)
            last = f"second/{sec.n_layers - 1}/"
            canvas = np.ascontiguousarray(sec.forward(scene["inputs_m1"], batch_size=n_agents).transpose(0, 2, 3, 1))
            cq = (np.float32(self.s[last + "a_delta"]), int(self.s[last + "a_zp"]))
            taps["canvas"] = canvas
        else:
            pcodes, canvas, cq = self.pfn_scatter(scene, n_agents)
            taps["pillar_code"], taps["canvas"] = pcodes, canvas
        return canvas, cq

    def forward(self, scene, taps=None):
        n_agents = len(scene["agent_modality_list"])
        taps = {} if taps is None else taps
        canvas, cq = self.canvas(scene, n_agents, taps)
        cat, cat_q = self.backbone(canvas, cq, taps)
        shr, shr_q = self.shrinker(cat, cat_q, taps)
        taps["shrinker_q"] = shr_q
        n, h, w, c = shr.shape
        if self.has_codebook:
            codes = self.encode(shr, shr_q)
            taps["codes"] = codes.reshape(-1, n, h, w)
            feats = self.decode(codes).reshape(n, h, w, c)
        else:
            if bool(self.s.get("meta/compress", False)):          # NaiveCompressor (naive_compress.py:5-35 under quant_block.py:1543-1570)
                for name in ("compressor.encoder.0", "compressor.decoder.0", "compressor.decoder.1"):
                    shr, shr_q = self.conv(name, shr, [(0, shr.shape[3], shr_q[0], shr_q[1])])
                    taps[name] = shr
            feats = (shr.astype(np.float32) - np.float32(shr_q[1])) * np.float32(shr_q[0])
        taps["features"] = feats
        fused = self.fuse(feats, scene["pairwise_t_matrix"], scene["record_len"])
        taps["fused"] = fused
        cls, reg, dr = self.heads(fused)
        out = {"cls_preds": cls, "reg_preds": reg, "dir_preds": dr, "preds_tensor": np.concatenate([cls, reg, dr], axis=1)}
        if bool(self.s["meta/supervise_single"]):
            s_cls, s_reg, s_dir = self.heads(feats, "_single")
            out.update({"cls_preds_single": s_cls, "reg_preds_single": s_reg, "dir_preds_single": s_dir})
        return out
