"""TEST INFRASTRUCTURE: the deployed W8A8 HEAL Pyramid-fusion path restated on the CPU, stage by stage.

Consumes ``quantv2x_amd.ptq_state.export_ptq_state`` of a calibrated ``QuantModel`` over ``heter_pyramid_collab_codebook_mc[_encdec]``
(data only) and a numpy scene.  Stage -> reference function:
  pfn_scatter   QuantPillarVFE / PointPillarScatter (as ``spec.Oracle``)
  agent_backbone ``QuantResNetBEVBackbone`` of ``QuantBasicBlock``s (quant_block.py:68-97, :398-459): conv1 -> codes; conv2 and the
                1x1 shortcut stay fp32 (disable_act_quant); out = quant(relu(conv2 + shortcut)) with the block's own quantizer
  encode/decode ``UMGMQuantizer.encode`` / ``decode`` with D = 64 (heter_pyramid_collab_codebook_mc_encdec.py:33-181)
  pyramid       ``QuantPyramidFusion.forward_collab`` (quant_block.py:489-533): ResNeXt ``QuantBottleneck`` levels (:100-131) on every
                agent's decoded map, 1x1 occupancy head per level, score = sigmoid(occ) + 1e-4, ``weighted_fuse``
                (pyramid_fuse.py:17-62), deblocks on the fused fp32 maps, concat
  shrink/heads  ``QuantDownsampleConv`` + the 1x1 heads
Integer sums are exact (``orc_convg``), fp32 dot products are ascending-k fmaf chains; see ``qv2x_oracle.c``."""
import ctypes

import numpy as np

from . import geometry
from .spec import Oracle, _cf, _f32, _i32, _p, lib


def decode_tables_d(state, levels, D):
    """decode(c0, c1, c2) = bias + T0[c0] + T1[c1] + T2[c2] for a D-wide codebook (``spec.decode_tables`` with D free)."""
    g = lambda l, n: state[f"codebook/{l}/{n}"].astype(np.float64)
    tables, const, chain = [None] * levels, np.zeros(D), np.eye(D)
    for l in range(levels):
        front = chain @ g(l, "restore_w")
        tables[l] = (front @ g(l, "dqhead_w") @ g(l, "codebook").T).T
        const = const + front @ g(l, "dqhead_b") + chain @ g(l, "restore_b")
        if l < levels - 1:
            const = const + front @ g(l, "side_b")
            chain = front @ g(l, "side_w")
    return np.stack(tables).astype(np.float32), const.astype(np.float32)


def sigmoid_lut(da, za):
    """score of every occupancy code: sigmoid((code - zp) * delta) + 1e-4, float64 then fp32 (the engine builds the same table)."""
    occ = ((np.arange(256, dtype=np.float32) - np.float32(za)) * np.float32(da)).astype(np.float32)
    sig = (1.0 / (1.0 + np.exp(-occ.astype(np.float64)))).astype(np.float32)
    return (sig + np.float32(1e-4)).astype(np.float32), occ


class OraclePyramid(Oracle):
    def __init__(self, state):
        self.s = state
        assert str(state["meta/fusion_method"]) == "pyramid"
        self.nx, self.ny, _ = (int(v) for v in state["meta/grid"])
        self.names = [str(n) for n in state["meta/module_names"]]
        self.layer_nums = [int(v) for v in state["meta/layer_nums"]]
        self.strides = [int(v) for v in state["meta/layer_strides"]]
        self.p_nums = [int(v) for v in state["meta/pyramid_layer_nums"]]
        self.p_strides = [int(v) for v in state["meta/pyramid_layer_strides"]]
        self.ups = [int(v) for v in state["meta/upsample_strides"]]
        self.has_codebook = bool(state["meta/has_codebook"])          # heter_pyramid_collab_mc (LiDAROnly/lidar_pyramid.yaml): no codebook
        if self.has_codebook:
            self.levels = int(state["meta/codebook_levels"])
            self.D = int(state["codebook/0/codebook"].shape[1])
            self.lut, self.lut_bias = decode_tables_d(state, self.levels, self.D)

    # ---- layers -----------------------------------------------------------------------------------------------------------
    def convg(self, name, x, xq, stride=1, f32_out=False, relu=True):
        """x u8 [N,H,W,Cin] with quantizer xq = (delta, zp).  Codes + (delta, zp) of the layer's quantizer, or the fp32 map."""
        p = self.q(name)
        code = np.ascontiguousarray(p["code"])
        cout, cg, k = code.shape[0], code.shape[1], code.shape[2]
        n, h, w, cin = x.shape
        groups = cin // cg
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        scale = (np.float32(xq[0]) * p["dw"].astype(np.float32)).astype(np.float32)
        out_u8 = np.zeros((n, ho, wo, cout), np.uint8) if not f32_out else None
        out_f = np.zeros((n, ho, wo, cout), np.float32) if f32_out else None
        lib().orc_convg(_p(np.ascontiguousarray(x)), n, h, w, cin, int(xq[1]), k, stride, groups, _p(code), _p(_i32(p["zw"])), cout,
                        _p(scale), _p(_f32(p["bias"])), 1 if f32_out else 0, 1 if relu else 0, _cf(p["da"]), _cf(p["za"]),
                        _p(out_u8) if out_u8 is not None else None, _p(out_f) if out_f is not None else None)
        if f32_out:
            assert p["a_off"], name
            return out_f
        assert not p["a_off"], name
        return out_u8, (p["da"], int(p["za"]))

    def dense_f32in(self, name, xf, s=1, out=None, out_c0=0, transposed=True):
        """A deblock (ConvTranspose2d, k = s) or a 1x1 convolution on an fp32 map [N,H,W,Cin]."""
        p = self.q(name)
        wdeq = _f32(self.dequant_weight(p))
        if not transposed:                                    # Conv2d 1x1 [Cout, Cin, 1, 1] -> the deconv layout [Cin, Cout, 1, 1]
            wdeq = _f32(wdeq.transpose(1, 0, 2, 3))
        n, h, w, cin = xf.shape
        cout = wdeq.shape[1]
        if out is None:
            out = np.zeros((n, h * s, w * s, cout), np.uint8)
        lib().orc_deconv_f32in(_p(_f32(xf)), n, h, w, cin, _p(wdeq), _p(_f32(p["bias"])), cout, s, 1, _cf(p["da"]), _cf(p["za"]),
                               _p(out), out.shape[3], out_c0)
        return out, (p["da"], int(p["za"]))

    def finish(self, block, y, res):
        da, za = np.float32(self.s[block + "/a_delta"]), np.float32(self.s[block + "/a_zp"])
        out = np.zeros(y.shape, np.uint8)
        lib().orc_add_relu_quant(_p(_f32(y)), _p(_f32(res)), ctypes.c_size_t(y.size), _cf(da), _cf(za), _p(out))
        return out, (da, int(za))

    @staticmethod
    def dequant(x, xq):
        return ((x.astype(np.int32) - int(xq[1])).astype(np.float32) * np.float32(xq[0])).astype(np.float32)

    def residual_block(self, name, x, xq, stride, convs, taps=None, x_f32=None):
        """One QuantBasicBlock (convs = ['conv1', 'conv2']) or QuantBottleneck (['conv1', 'conv2', 'conv3']).  ``x_f32``: the block
        input when it is not on a quantizer grid (the decoded feature into the first pyramid block)."""
        has_ds = (name + ".downsample/w_code") in self.s
        strided = "conv1" if len(convs) == 2 else "conv2"          # BasicBlock strides conv1, Bottleneck conv2 (resblock.py:41, :104)
        cur, cq = x, xq
        for c in convs[:-1]:
            st = stride if c == strided else 1
            if cur is None:
                cur, cq = self.dense_f32in(f"{name}.{c}", x_f32, transposed=False)
            else:
                cur, cq = self.convg(f"{name}.{c}", cur, cq, stride=st)
            if taps is not None:
                taps[f"{name}.{c}"] = cur
        y = self.convg(f"{name}.{convs[-1]}", cur, cq, f32_out=True)
        if has_ds:
            if x is None:
                raise NotImplementedError("a 1x1 shortcut on an fp32 input")
            res = self.convg(f"{name}.downsample", x, xq, stride=stride, f32_out=True)
        else:
            res = x_f32 if x is None else self.dequant(x, xq)
        out, oq = self.finish(name, y, res)
        if taps is not None:
            taps[name] = out
        return out, oq

    # ---- stages -----------------------------------------------------------------------------------------------------------
    def agent_backbone(self, canvas, cq, taps=None):
        x, xq = canvas, cq
        for b in range(self.layer_nums[0]):
            x, xq = self.residual_block(f"backbone_m1.resnet.layer0.{b}", x, xq, self.strides[0] if b == 0 else 1, ["conv1", "conv2"], taps)
        return x, xq

    def encode(self, feat, feat_q):
        rows = self.dequant(feat, feat_q).reshape(-1, feat.shape[-1])
        return self.encode_rows(rows)

    def encode_rows(self, rows, want_gaps=False):
        L, R, keep = self.levels, rows.shape[0], []

        def arr(tag):
            ptrs = (ctypes.c_void_p * L)()
            for l in range(L):
                key = f"codebook/{l}/{tag}"
                if key in self.s:
                    a = _f32(self.s[key]); keep.append(a); ptrs[l] = a.ctypes.data
                else:
                    ptrs[l] = None
            return ptrs
        S = int(self.s.get("meta/codebook_segs", 1))                    # seg_num: the state holds the extended codebook [S * kc][D]
        kc = self.s["codebook/0/codebook"].shape[0] // S
        codes = np.zeros((L * S, R), np.uint8)
        gaps = np.zeros((L * S, R), np.float32) if want_gaps else None
        lib().orc_codebook_encode_seg(_p(_f32(rows)), R, L, kc, self.D, S, arr("stage_w"), arr("stage_b"), arr("qhead_w"), arr("qhead_b"),
                                      arr("lhead_w"), arr("lhead_b"), arr("codebook"), _p(codes), _p(gaps) if want_gaps else None)
        return (codes, gaps) if want_gaps else codes

    def decode(self, codes):
        P, R = codes.shape                                                # planes = levels * seg_num
        out = np.zeros((R, self.D), np.float32)
        kc = self.lut.shape[0] * self.lut.shape[1] // P
        lib().orc_decode_lut_d(_p(np.ascontiguousarray(codes)), R, P, kc, self.D, _p(_f32(self.lut)), _p(_f32(self.lut_bias)), _p(out))
        return out

    def occupancy(self, lvl, x, xq):
        """occupancy codes, fp32 occupancy map and score of one level: [N,h,w,1] each."""
        code, oq = self.convg(f"pyramid_backbone.single_head_{lvl}", x, xq, relu=False)
        lut, occ = sigmoid_lut(*oq)
        return code, occ[code], lut[code]

    def pyramid(self, feats, pairwise_t, record_len, taps=None, codes_in=None):
        """feats f32 [sum_N, h, w, D] (decoded) -- or, without a codebook, ``codes_in = (x, xq)``: the agents' activation codes --
        -> concat of the deblocks' codes [B, h, w, 384] with its three quantizers."""
        H, W = (float(v) for v in self.s["meta/HW_metres"])
        affine = geometry.normalize_pairwise_tfm(np.asarray(pairwise_t), H, W, float(self.s["meta/discrete_ratio"]))
        x, xq, x_f32 = (None, None, feats) if codes_in is None else (codes_in[0], codes_in[1], None)
        cat, cat_q, c0, occs = None, [], 0, []
        total = sum(self.s[f"pyramid_backbone.deblocks.{l}.0/w_code"].shape[1] for l in range(len(self.ups)))
        for lvl in range(len(self.p_nums)):
            for b in range(self.p_nums[lvl]):
                x, xq = self.residual_block(f"pyramid_backbone.resnet.layer{lvl}.{b}", x, xq, self.p_strides[lvl] if b == 0 else 1,
                                            ["conv1", "conv2", "conv3"], taps, x_f32=x_f32)
                x_f32 = None
            ocode, occ, score = self.occupancy(lvl, x, xq)
            occs.append(occ.transpose(0, 3, 1, 2))
            fx = self.dequant(x, xq)
            fused, start = [], 0
            for bi, n in enumerate(int(v) for v in record_len):
                fused.append(geometry.weighted_fuse(fx[start:start + n], score[start:start + n], affine[bi], n))
                start += n
            fused = np.stack(fused)
            if taps is not None:
                taps[f"occ_code{lvl}"], taps[f"score{lvl}"], taps[f"fused{lvl}"] = ocode, score, fused
            s = self.ups[lvl]
            name = f"pyramid_backbone.deblocks.{lvl}.0"
            cup = self.s[name + "/w_code"].shape[1]
            if cat is None:
                cat = np.zeros((fused.shape[0], fused.shape[1] * s, fused.shape[2] * s, total), np.uint8)
            _, oq = self.dense_f32in(name, fused, s, cat, c0)
            cat_q.append((c0, cup, oq[0], oq[1]))
            if taps is not None:
                taps[name] = cat[..., c0:c0 + cup]
            c0 += cup
        return cat, cat_q, occs

    def shrink(self, cat, cat_q, taps=None):
        x, xq = self.conv("shrink_conv.layers.0.double_conv.0", cat, cat_q)
        y, yq = self.conv("shrink_conv.layers.0.double_conv.1", x, [(0, x.shape[3], xq[0], xq[1])])
        if taps is not None:
            taps["shrink_conv.layers.0.double_conv.0"], taps["shrink_conv.layers.0.double_conv.1"] = x, y
        return y, yq

    # ---- whole path: what an agent runs before the link, what the ego runs on the received codes ------------------------------
    def encode_features(self, scene, taps=None):
        n_agents = len(scene["agent_modality_list"])
        taps = {} if taps is None else taps
        pcodes, canvas, cq = self.pfn_scatter(scene, n_agents)
        taps["pillar_code"], taps["canvas"] = pcodes, canvas
        x, xq = self.agent_backbone(canvas, cq, taps)
        taps["agent_feature"], taps["agent_feature_q"] = x, xq
        n, h, w, _ = x.shape
        codes = self.encode(x, xq)
        taps["codes"] = codes.reshape(-1, n, h, w)
        return codes, (n, h, w)

    def decode_features(self, codes, shape, scene, taps=None):
        taps = {} if taps is None else taps
        n, h, w = shape
        feats = self.decode(codes).reshape(n, h, w, self.D)
        taps["features"] = feats
        cat, cat_q, occs = self.pyramid(feats, scene["pairwise_t_matrix"], scene["record_len"], taps)
        taps["cat"] = cat
        shr, shr_q = self.shrink(cat, cat_q, taps)
        cls, reg, dr = self.heads(self.dequant(shr, shr_q))
        return {"cls_preds": cls, "reg_preds": reg, "dir_preds": dr, "preds_tensor": np.concatenate([cls, reg, dr], axis=1),
                "occ_single_list": occs}

    def forward(self, scene, taps=None):
        if self.has_codebook:
            codes, shape = self.encode_features(scene, taps)
            return self.decode_features(codes, shape, scene, taps)
        taps = {} if taps is None else taps
        pcodes, canvas, cq = self.pfn_scatter(scene, len(scene["agent_modality_list"]))
        taps["pillar_code"], taps["canvas"] = pcodes, canvas
        x, xq = self.agent_backbone(canvas, cq, taps)
        cat, cat_q, occs = self.pyramid(None, scene["pairwise_t_matrix"], scene["record_len"], taps, codes_in=(x, xq))
        taps["cat"] = cat
        shr, shr_q = self.shrink(cat, cat_q, taps)
        cls, reg, dr = self.heads(self.dequant(shr, shr_q))
        return {"cls_preds": cls, "reg_preds": reg, "dir_preds": dr, "preds_tensor": np.concatenate([cls, reg, dr], axis=1),
                "occ_single_list": occs}
