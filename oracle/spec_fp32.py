"""TEST INFRASTRUCTURE: the un-quantized (fp32) hot path restated on the CPU in the operation order of ``csrc/fp32_path.hip``.

Consumes the numpy state of ``quantv2x_amd.engine_fp32.export_fp32_state`` (data only).  a1-a4 are ``orc_pfn_f32`` /
``orc_gemm_f32`` (one fmaf chain per output, K in groups of 8 walked k0 k4 k1 k5 k2 k6 k3 k7); the codebook, the geometry and the
heads are the functions the W8A8 oracle uses (``spec.Oracle``), with the head output quantizers off.  Reference: the plain fp32
forward of ``heter_baseline_collab_codebook.py:71-169`` with the deterministic encode -> decode pair."""
import ctypes

import numpy as np

from . import spec
from .spec import _cf, _f32, _i32, _p, lib


class OracleFp32(spec.Oracle):
    def __init__(self, state):
        self.s = state
        self.nx, self.ny, _ = (int(v) for v in state["meta/grid"])
        self.layer_nums = [int(v) for v in state["meta/layer_nums"]]
        self.strides = [int(v) for v in state["meta/layer_strides"]]
        self.ups = [int(v) for v in state["meta/upsample_strides"]]
        self.has_codebook = bool(state["meta/has_codebook"])
        if self.has_codebook:
            self.lut, self.lut_bias = spec.decode_tables(state, int(state["meta/codebook_levels"]))

    def gemm(self, name, x, stride, deconv=False, out=None, out_c0=0, cin0=0, cin=None):
        w, b = self.s[name + "/w"], _f32(self.s[name + "/bias"])
        n, h, ww, ct = x.shape
        if deconv:
            cout, s = w.shape[1], w.shape[2]
            wmat = _f32(w.transpose(2, 3, 1, 0).reshape(-1, w.shape[0]))
            ho, wo = h * s, ww * s
        else:
            cout = w.shape[0]
            wmat = _f32(w.transpose(0, 2, 3, 1).reshape(cout, -1))
            ho, wo = (h + 2 - 3) // stride + 1, (ww + 2 - 3) // stride + 1
        cin = (w.shape[0] if deconv else w.shape[1]) if cin is None else cin
        if out is None:
            out = np.zeros((n, ho, wo, cout), np.float32)
        lib().orc_gemm_f32(_p(_f32(x)), n, h, ww, ct, cin0, cin, stride, cout, 1 if deconv else 0, _p(wmat), _p(b), 1,
                           _p(out), out.shape[3], out_c0)
        return out

    def forward(self, scene, taps=None):
        s = self.s
        taps = {} if taps is None else taps
        n = len(scene["agent_modality_list"])
        inp = scene["inputs_m1"]
        vf, co, npt = _f32(inp["voxel_features"]), _i32(inp["voxel_coords"]), _i32(inp["voxel_num_points"])
        feats = np.zeros((vf.shape[0], 64), np.float32)
        lib().orc_pfn_f32(_p(vf), _p(co), _p(npt), vf.shape[0], vf.shape[1], _p(_f32(s["pfn/w"])), _p(_f32(s["pfn/bias"])),
                          _p(_f32(s["meta/voxel"])), _p(_f32(s["meta/offset"])), _p(feats))
        canvas = np.zeros((n, self.ny, self.nx, 64), np.float32)
        for m in range(vf.shape[0]):                           # later pillars overwrite earlier ones, as the scatter does
            b, z, y, x = co[m]
            canvas[b, y, z + x] = feats[m]
        taps["canvas"] = canvas
        x, cat, c0 = canvas, None, 0
        for lvl in range(len(self.layer_nums)):
            for i in range(self.layer_nums[lvl] + 1):
                name = f"backbone_m1.blocks.{lvl}.{i + 1}"
                x = self.gemm(name, x, self.strides[lvl] if i == 0 else 1)
                taps[name] = x
            name = f"backbone_m1.deblocks.{lvl}.0"
            cup, up = s[name + "/w"].shape[1], self.ups[lvl]
            if cat is None:
                total = sum(s[f"backbone_m1.deblocks.{l}.0/w"].shape[1] for l in range(len(self.ups)))
                cat = np.zeros((n, x.shape[1] * up, x.shape[2] * up, total), np.float32)
            self.gemm(name, x, up, deconv=True, out=cat, out_c0=c0)
            c0 += cup
        taps["cat"] = cat
        s0 = self.gemm("shrinker_m1.layers.0.double_conv.0", cat, 1)
        s1 = self.gemm("shrinker_m1.layers.0.double_conv.1", s0, 1)
        taps["shrinker_m1.layers.0.double_conv.0"], taps["shrinker_m1.layers.0.double_conv.1"] = s0, s1
        _, h, w, c = s1.shape
        if self.has_codebook:
            codes = self.encode_rows(s1.reshape(-1, c))
            taps["codes"] = codes.reshape(-1, n, h, w)
            feats2d = self.decode(codes).reshape(n, h, w, c)
        else:
            feats2d = s1
        taps["features"] = feats2d
        fused = self.fuse(feats2d, scene["pairwise_t_matrix"], scene["record_len"])
        taps["fused"] = fused
        cls, reg, dr = self.heads(fused)
        out = {"cls_preds": cls, "reg_preds": reg, "dir_preds": dr, "preds_tensor": np.concatenate([cls, reg, dr], axis=1)}
        if bool(s["meta/supervise_single"]):
            s_cls, s_reg, s_dir = self.heads(feats2d, "_single")
            out.update({"cls_preds_single": s_cls, "reg_preds_single": s_reg, "dir_preds_single": s_dir})
        return out

    def heads(self, fused, suffix=""):
        b, h, w, c = fused.shape
        rows = _f32(fused.reshape(-1, c))
        outs = []
        for head in ("cls_head", "reg_head", "dir_head"):
            wm, bias = _f32(self.s[head + suffix + "/w"]), _f32(self.s[head + suffix + "/bias"])
            o = np.zeros((rows.shape[0], wm.shape[0]), np.float32)
            lib().orc_heads(_p(rows), rows.shape[0], c, _p(wm), _p(bias), wm.shape[0], 0, _cf(1.0), _cf(0.0), _p(o))
            outs.append(o.reshape(b, h, w, -1).transpose(0, 3, 1, 2))
        return outs


class OraclePyramidFp32(OracleFp32):
    """The un-quantized HEAL Pyramid model in the operation order of ``quantv2x_amd/engine_pyramid_fp32.py``: every convolution one
    ``orc_gemm_f32`` (1x1 = the deconvolution form with stride 1; the strided 1x1 shortcut = the 3x3 form with only the centre tap set;
    a grouped 3x3 = one dense block-diagonal GEMM per 64-channel slab), ``relu(branch + shortcut)`` in numpy, codebook D = 64.
    Reference: ``heter_pyramid_collab_codebook_mc_encdec.py:33-181`` in fp32 (BatchNorm folded in float64 by the state export)."""

    def __init__(self, state):
        self.s = state
        self.nx, self.ny, _ = (int(v) for v in state["meta/grid"])
        self.has_codebook = bool(state["meta/has_codebook"])
        self.p_nums = [int(v) for v in state["meta/pyramid_layer_nums"]]
        self.p_strides = [int(v) for v in state["meta/pyramid_layer_strides"]]
        self.ups = [int(v) for v in state["meta/upsample_strides"]]
        if self.has_codebook:
            from .spec_pyramid import decode_tables_d
            self.levels, self.D = int(state["meta/codebook_levels"]), int(state["codebook/0/codebook"].shape[1])
            self.lut, self.lut_bias = decode_tables_d(state, self.levels, self.D)

    @staticmethod
    def _run(x, wmat, bias, stride, cout, deconv, relu, cin0=0, cin=None, out=None, out_c0=0):
        n, h, w, ct = x.shape
        cin = ct if cin is None else cin
        ho, wo = (h * stride, w * stride) if deconv else ((h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1)
        if out is None:
            out = np.zeros((n, ho, wo, cout), np.float32)
        lib().orc_gemm_f32(_p(_f32(x)), n, h, w, ct, cin0, cin, stride, cout, 1 if deconv else 0, _p(_f32(wmat)), _p(_f32(bias)), 1 if relu else 0,
                           _p(out), out.shape[3], out_c0)
        return out

    def conv3x3(self, name, x, stride, relu=True):
        w = self.s[name + "/w"]
        return self._run(x, w.transpose(0, 2, 3, 1).reshape(w.shape[0], -1), self.s[name + "/bias"], stride, w.shape[0], False, relu)

    def conv1x1(self, name, x, relu=True):
        w = self.s[name + "/w"]
        return self._run(x, w.reshape(w.shape[0], -1), self.s[name + "/bias"], 1, w.shape[0], True, relu)

    def shortcut(self, name, x, stride):
        w = self.s[name + "/w"]
        full = np.zeros((w.shape[0], 9, w.shape[1]), np.float32)
        full[:, 4, :] = w.reshape(w.shape[0], w.shape[1])                     # the centre tap of a 3x3 window = a 1x1 convolution
        return self._run(x, full.reshape(w.shape[0], -1), self.s[name + "/bias"], stride, w.shape[0], False, False)

    def grouped3x3(self, name, x, stride):
        w, bias = self.s[name + "/w"], self.s[name + "/bias"]                  # [c, cg, 3, 3]
        c, cg = w.shape[:2]
        out = None
        for s0 in range(0, c, 64):
            m = np.zeros((64, 9, 64), np.float32)
            for co in range(64):
                g0 = (co // cg) * cg
                m[co, :, g0:g0 + cg] = w[s0 + co].reshape(cg, 9).T
            if out is None:
                n, h, ww, _ = x.shape
                out = np.zeros((n, (h + 2 - 3) // stride + 1, (ww + 2 - 3) // stride + 1, c), np.float32)
            self._run(x, m.reshape(64, -1), bias[s0:s0 + 64], stride, 64, False, True, cin0=s0, cin=64, out=out, out_c0=s0)
        return out

    def block(self, name, x, stride, bottleneck, taps):
        if bottleneck:
            y = self.conv1x1(name + ".conv3", self.grouped3x3(name + ".conv2", self.conv1x1(name + ".conv1", x), stride), relu=False)
        else:
            y = self.conv3x3(name + ".conv2", self.conv3x3(name + ".conv1", x, stride), 1, relu=False)
        res = self.shortcut(name + ".downsample", x, stride) if (name + ".downsample/w") in self.s else x
        out = np.maximum(y + res, np.float32(0)).astype(np.float32)
        taps[name] = out
        return out

    def agent_features(self, scene, taps):
        s = self.s
        n = len(scene["agent_modality_list"])
        inp = scene["inputs_m1"]
        vf, co, npt = _f32(inp["voxel_features"]), _i32(inp["voxel_coords"]), _i32(inp["voxel_num_points"])
        feats = np.zeros((vf.shape[0], 64), np.float32)
        lib().orc_pfn_f32(_p(vf), _p(co), _p(npt), vf.shape[0], vf.shape[1], _p(_f32(s["pfn/w"])), _p(_f32(s["pfn/bias"])),
                          _p(_f32(s["meta/voxel"])), _p(_f32(s["meta/offset"])), _p(feats))
        canvas = np.zeros((n, self.ny, self.nx, 64), np.float32)
        for m in range(vf.shape[0]):
            b, z, y, x = co[m]
            canvas[b, y, z + x] = feats[m]
        x = canvas
        for b in range(int(s["meta/layer_nums"][0])):
            x = self.block(f"backbone_m1.resnet.layer0.{b}", x, int(s["meta/layer_strides"][0]) if b == 0 else 1, False, taps)
        return x

    def pyramid_and_heads(self, x, scene, taps):
        from . import geometry
        s = self.s
        H, W = (float(v) for v in s["meta/HW_metres"])
        affine = geometry.normalize_pairwise_tfm(np.asarray(scene["pairwise_t_matrix"]), H, W, float(s["meta/discrete_ratio"]))
        lens = [int(v) for v in scene["record_len"]]
        cat, c0, occs = None, 0, []
        for lvl, nb in enumerate(self.p_nums):
            for b in range(nb):
                x = self.block(f"pyramid_backbone.resnet.layer{lvl}.{b}", x, self.p_strides[lvl] if b == 0 else 1, True, taps)
            hw = s[f"pyramid_backbone.single_head_{lvl}/w"]
            wm = np.zeros((64, hw.shape[1]), np.float32); wm[0] = hw.reshape(-1)
            bm = np.zeros(64, np.float32); bm[0] = s[f"pyramid_backbone.single_head_{lvl}/bias"][0]
            occ = self._run(x, wm, bm, 1, 64, True, False)[..., :1]
            score = (1.0 / (1.0 + np.exp(-occ.astype(np.float32)))).astype(np.float32) + np.float32(1e-4)
            occs.append(occ.transpose(0, 3, 1, 2))
            fused, start = [], 0
            for bi, na in enumerate(lens):
                fused.append(geometry.weighted_fuse(x[start:start + na], score[start:start + na], affine[bi], na))
                start += na
            fused = np.stack(fused)
            taps[f"score{lvl}"], taps[f"fused{lvl}"] = score, fused
            name = f"pyramid_backbone.deblocks.{lvl}.0"
            w = s[name + "/w"]
            up = self.ups[lvl]
            if cat is None:
                total = sum(s[f"pyramid_backbone.deblocks.{l}.0/w"].shape[1] for l in range(len(self.ups)))
                cat = np.zeros((fused.shape[0], fused.shape[1] * up, fused.shape[2] * up, total), np.float32)
            self._run(fused, w.transpose(2, 3, 1, 0).reshape(-1, w.shape[0]), s[name + "/bias"], up, w.shape[1], True, True, out=cat, out_c0=c0)
            c0 += w.shape[1]
        taps["cat"] = cat
        s1 = self.conv3x3("shrink_conv.layers.0.double_conv.1", self.conv3x3("shrink_conv.layers.0.double_conv.0", cat, 1), 1)
        taps["shrink_conv.layers.0.double_conv.1"] = s1
        cls, reg, dr = self.heads(s1)
        return {"cls_preds": cls, "reg_preds": reg, "dir_preds": dr, "preds_tensor": np.concatenate([cls, reg, dr], axis=1), "occ_single_list": occs}

    def forward(self, scene, taps=None):
        taps = {} if taps is None else taps
        x = self.agent_features(scene, taps)
        if self.has_codebook:
            n, h, w, c = x.shape
            L, R, keep = self.levels, n * h * w, []

            def arr(tag):
                ptrs = (ctypes.c_void_p * L)()
                for l in range(L):
                    key = f"codebook/{l}/{tag}"
                    if key in self.s:
                        a = _f32(self.s[key]); keep.append(a); ptrs[l] = a.ctypes.data
                    else:
                        ptrs[l] = None
                return ptrs
            S = int(self.s.get("meta/codebook_segs", 1))                # seg_num: the state holds the extended codebook [S * kc][D]
            kc = self.s["codebook/0/codebook"].shape[0] // S
            codes = np.zeros((L * S, R), np.uint8)                       # planes
            lib().orc_codebook_encode_seg(_p(_f32(x.reshape(R, c))), R, L, kc, self.D, S, arr("stage_w"), arr("stage_b"), arr("qhead_w"),
                                          arr("qhead_b"), arr("lhead_w"), arr("lhead_b"), arr("codebook"), _p(codes), None)
            taps["codes"] = codes.reshape(L * S, n, h, w)
            out = np.zeros((R, self.D), np.float32)
            lib().orc_decode_lut_d(_p(codes), R, L * S, kc, self.D, _p(_f32(self.lut)), _p(_f32(self.lut_bias)), _p(out))
            x = out.reshape(n, h, w, c)
            taps["features"] = x
        return self.pyramid_and_heads(x, scene, taps)
