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
