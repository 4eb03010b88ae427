"""The quantized SECOND encoder on the HIP path: ``QuantSECOND.forward`` (reference
``opencood/quant/quant_block.py:1037-1078``) = MeanVFE -> the twelve sparse 3-D convolutions of ``VoxelBackBone8x`` -> ``HeightCompression``,
compiled from the frozen W8A8 state ``ptq_state.export_second_state`` extracts.

Everything runs through ``csrc/sparse_conv.hip`` (C ABI ``qv2x_mean_vfe_f32`` / ``qv2x_sp_*``): row counts stay on the device, so one frame
is a fixed sequence of launches with no host round trip.  The output is the padded i8 BEV map ``[agents][H/8 + 2][W/8 + 2][C * D]`` the 2-D
int8 layers take (``code - 128``, cells without a site = the real value 0), plus its quantizer ``out_q``.

Capacities: a strided 3x3x3 convolution can activate up to 8 outputs per input site, so a level holds ``min(8 * rows below, cells)`` rows;
at 288 GB per GPU the dense index volumes (369 MB per agent at 0.1 m over the V2X-Real range) and these worst-case row buffers are noise.
There is no CPU fallback: without ``libqv2x.so`` construction fails.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import lib as L
from .engine import _dev


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _SpLayer:
    """Device image of one sparse convolution."""

    def __init__(self, state, i, in_q, dev):
        p = f"second/{i}/"
        code = state[p + "w_code"].astype(np.int64)                     # [K, Cin, Cout]
        self.K, self.ci, self.co = code.shape
        g = [int(v) for v in state[p + "geom"]]
        self.subm, self.k, self.s, self.p = g[0], g[1:4], g[4:7], g[7:10]
        self.out_q = (float(state[p + "a_delta"]), float(state[p + "a_zp"]))
        self.cop = _pad32(self.co)
        zw = state[p + "w_zp"].astype(np.int64)
        gq, hq = np.zeros(self.cop, np.float32), np.zeros(self.cop, np.float32)
        gq[: self.co], hq[: self.co] = state[p + "bn_g"], state[p + "bn_h"]
        self.bn_g, self.bn_h = _dev(gq, dev), _dev(hq, dev)
        self.f32_in = in_q is None
        if self.f32_in:
            if self.ci != 4 or self.co != 16:
                raise NotImplementedError("the fp32-input sparse layer is conv_input: 4 -> 16 channels")
            wd = ((code - zw[None, None, :]).astype(np.float32) * state[p + "w_delta"].astype(np.float32)[None, None, :]).astype(np.float32)
            self.w = _dev(np.ascontiguousarray(wd), dev)
            self.cip = 4
            return
        self.cip = _pad32(self.ci)
        ks, nt = self.cip // 32, self.cop // 32
        ws = np.zeros((self.K, self.cip, self.cop), np.int64)
        ws[:, : self.ci, : self.co] = code - 128
        ws[:, self.ci:, : self.co] = (zw - 128)[None, None, :]         # padded input channels: w - zp_w = 0
        aw = np.zeros(self.cop, np.int64)
        aw[: self.co] = 128 - zw
        ax = 128 - int(in_q[1])
        corr = ax * ws.sum(axis=(0, 1)) + self.K * self.cip * ax * aw
        corr[self.co:] = 0
        lane = np.arange(64)
        cin_idx = (np.arange(ks)[:, None, None] * 32 + (lane >> 5)[None, :, None] * 16 + np.arange(16)[None, None, :])     # [ks, lane, 16]
        cout_idx = (np.arange(nt)[:, None] * 32 + (lane & 31)[None, :])                                                  # [nt, lane]
        frag = ws[:, cin_idx[:, None, :, :], cout_idx[None, :, :, None]]                                                 # [K, ks, nt, lane, 16]
        self.w = _dev(np.ascontiguousarray(frag.astype(np.int8)), dev)
        sc = np.zeros(self.cop, np.float32)
        sc[: self.co] = (state[p + "w_delta"].astype(np.float32) * np.float32(in_q[0])).astype(np.float32)
        self.scale, self.corr, self.aw = _dev(sc, dev), _dev(corr.astype(np.int32), dev), _dev(aw.astype(np.int32), dev)


class _Level:
    """One resolution of the sparse volume: coordinates, row count, dense index volume."""

    def __init__(self, shape, agents, cap, dev, own_coords=True):
        self.shape, self.cap = [int(v) for v in shape], int(cap)
        self.volume = torch.full((agents * self.shape[0] * self.shape[1] * self.shape[2],), -1, dtype=torch.int32, device=dev)
        self.coords = torch.zeros((self.cap, 4), dtype=torch.int32, device=dev) if own_coords else None
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scan_ws = torch.zeros((self.volume.numel() + 1023) // 1024 * 4 + 16, dtype=torch.uint8, device=dev)      # qv2x_sp_out_sites_workspace_bytes


class DeployedSecondEncoder:
    def __init__(self, state: Dict[str, np.ndarray], device="cuda:0", agents: int = 1, max_voxels: int = 70000, max_points: int = 5,
                 site_caps: Optional[List[int]] = None):
        self.lib = L.load()
        self.dev = torch.device(device)
        self.agents, self.max_points = int(agents), int(max_points)
        n = int(state["second/n_layers"])
        self.layers: List[_SpLayer] = []
        q = None
        for i in range(n):
            self.layers.append(_SpLayer(state, i, q, self.dev))
            q = self.layers[-1].out_q
        self.out_q = q
        # levels: level 0 = the voxelizer's sites; every SparseConv3d opens the next one
        shape = [int(v) for v in state["second/sparse_shape"]]
        cap = self.agents * int(max_voxels)
        self.levels = [_Level(shape, self.agents, cap, self.dev, own_coords=False)]
        self.level_of: List[int] = []                                   # per layer: level of its OUTPUT
        for li, ly in enumerate(self.layers):
            if not ly.subm:
                shape = [(d + 2 * p - k) // s + 1 for d, p, k, s in zip(shape, ly.p, ly.k, ly.s)]
                if min(shape) <= 0:
                    raise ValueError(f"sparse layer {li}: the volume {self.levels[-1].shape} is too small for its window")
                cells = self.agents * shape[0] * shape[1] * shape[2]
                cap = min(8 * cap, cells) if ly.K == 27 else min(cap, cells)
                if site_caps is not None:
                    cap = min(cap, int(site_caps[len(self.levels) - 1]))
                self.levels.append(_Level(shape, self.agents, cap, self.dev))
            self.level_of.append(len(self.levels) - 1)
        # feature rows per layer output (+ the fill row), rulebooks per (input level, output level, window)
        self.feats: List[torch.Tensor] = []
        for li, ly in enumerate(self.layers):
            lv = self.levels[self.level_of[li]]
            f = torch.empty((lv.cap + 1, ly.cop), dtype=torch.int8, device=self.dev)
            f[lv.cap] = int(ly.out_q[1]) - 128
            self.feats.append(f)
        self.mean = torch.zeros((self.levels[0].cap, 4), dtype=torch.float32, device=self.dev)
        self.nbr: Dict[tuple, torch.Tensor] = {}
        for li, ly in enumerate(self.layers):
            key = self._rb_key(li)
            if key not in self.nbr:
                self.nbr[key] = torch.empty((ly.K, self.levels[self.level_of[li]].cap), dtype=torch.int32, device=self.dev)
        last, lv = self.layers[-1], self.levels[-1]
        self.bev_channels = last.co * lv.shape[0]
        self.bev = torch.empty((self.agents, lv.shape[1] + 2, lv.shape[2] + 2, self.bev_channels), dtype=torch.int8, device=self.dev)

    def _rb_key(self, li):
        ly = self.layers[li]
        src = self.level_of[li] if ly.subm else self.level_of[li] - 1
        return (src, self.level_of[li], tuple(ly.k), tuple(ly.s), tuple(ly.p))

    def _desc(self, li) -> L.SpconvDesc:
        ly = self.layers[li]
        dst = self.levels[self.level_of[li]]
        src = dst if ly.subm else self.levels[self.level_of[li] - 1]
        d = L.SpconvDesc()
        d.subm = ly.subm
        for a in range(3):
            d.k[a], d.s[a], d.p[a] = ly.k[a], ly.s[a], ly.p[a]
            d.in_shape[a], d.out_shape[a] = src.shape[a], dst.shape[a]
        d.agents, d.cin, d.cout, d.cap_in, d.cap_out = self.agents, ly.cip, ly.cop, src.cap, dst.cap
        d.out_delta, d.out_zp = ly.out_q
        return d

    @torch.no_grad()
    def forward(self, inputs: Dict[str, torch.Tensor], taps: Optional[dict] = None, n_voxels: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``inputs``: voxel_features f32 [M, max_points, 4], voxel_coords i32 [M, 4] = (agent, z, y, x), voxel_num_points i32 [M]; ``n_voxels``
        (device int32 [1], e.g. the voxelizer's count) defaults to M.  Returns the padded i8 BEV map (a buffer the engine owns)."""
        vf, co, npnt = inputs["voxel_features"], inputs["voxel_coords"], inputs["voxel_num_points"]
        lv0 = self.levels[0]
        if vf.dtype != torch.float32 or co.dtype != torch.int32 or npnt.dtype != torch.int32:
            raise TypeError("voxel_features f32, voxel_coords / voxel_num_points int32")
        if vf.shape[0] > lv0.cap or vf.shape[1] != self.max_points or vf.shape[2] != 4:
            raise ValueError(f"voxel_features {tuple(vf.shape)}: at most {lv0.cap} voxels of {self.max_points} x 4")
        vf, co, npnt = vf.contiguous(), co.contiguous(), npnt.contiguous()
        if vf.shape[0] == 0:                                          # an empty sweep: one dummy row that the row count (0) hides
            vf, co, npnt = vf.new_zeros((1, self.max_points, 4)), co.new_zeros((1, 4)), npnt.new_zeros((1,))
            n_voxels = torch.zeros(1, dtype=torch.int32, device=vf.device)
        if n_voxels is None:
            lv0.count.fill_(vf.shape[0])
        else:
            lv0.count.copy_(n_voxels.reshape(1))
        lv0.coords = co
        st = L.current_stream()
        lib = self.lib
        # level 0 lives in the caller's buffers (M rows); the kernels never index past the device-side row count
        L.check(lib.qv2x_mean_vfe_f32(L.ptr(vf), L.ptr(npnt), L.ptr(lv0.count), lv0.cap, self.max_points, L.ptr(self.mean), st), "qv2x_mean_vfe_f32")
        L.check(lib.qv2x_sp_index_scatter(L.ptr(co), L.ptr(lv0.count), lv0.cap, self.agents, *lv0.shape, L.ptr(lv0.volume), 1, st), "qv2x_sp_index_scatter")
        built = set()
        x = self.mean
        for li, ly in enumerate(self.layers):
            d = self._desc(li)
            dst = self.levels[self.level_of[li]]
            src = dst if ly.subm else self.levels[self.level_of[li] - 1]
            key = self._rb_key(li)
            nbr = self.nbr[key]
            if not ly.subm:
                L.check(lib.qv2x_sp_out_sites(C.byref(d), L.ptr(src.coords), L.ptr(src.count), L.ptr(dst.volume), L.ptr(dst.coords), L.ptr(dst.count),
                                              L.ptr(dst.scan_ws), dst.scan_ws.numel(), st), f"qv2x_sp_out_sites[{li}]")
            if key not in built:                                      # sub-manifold layers of one `indice_key` share the rulebook
                L.check(lib.qv2x_sp_rulebook(C.byref(d), L.ptr(dst.coords), L.ptr(dst.count), L.ptr(src.volume), L.ptr(nbr), st), f"qv2x_sp_rulebook[{li}]")
                built.add(key)
            out = self.feats[li]
            if ly.f32_in:
                L.check(lib.qv2x_sp_conv_f32in(C.byref(d), L.ptr(x), L.ptr(nbr), L.ptr(dst.count), L.ptr(ly.w), L.ptr(ly.bn_g), L.ptr(ly.bn_h), L.ptr(out), st),
                        "qv2x_sp_conv_f32in")
            else:
                L.check(lib.qv2x_sp_conv_i8(C.byref(d), L.ptr(x), L.ptr(nbr), L.ptr(dst.count), L.ptr(ly.w), L.ptr(ly.scale), L.ptr(ly.corr), L.ptr(ly.aw),
                                            L.ptr(ly.bn_g), L.ptr(ly.bn_h), L.ptr(out), st), f"qv2x_sp_conv_i8[{li}]")
            if taps is not None:
                taps[f"second/{li}"] = (out, dst.coords, dst.count, ly.co, list(dst.shape))
            x = out
        last, lv = self.layers[-1], self.levels[-1]
        L.check(lib.qv2x_sp_to_bev_i8(L.ptr(x), L.ptr(lv.coords), L.ptr(lv.count), lv.cap, last.co, last.cop, self.agents, *lv.shape,
                                      int(last.out_q[1]) - 128, L.ptr(self.bev), st), "qv2x_sp_to_bev_i8")
        # leave every index volume clean for the next frame: undo exactly the cells this frame set
        for i, l in enumerate(self.levels):
            L.check(lib.qv2x_sp_index_scatter(L.ptr(l.coords), L.ptr(l.count), l.cap, self.agents, *l.shape, L.ptr(l.volume), 0, st),
                    "qv2x_sp_index_scatter (undo)")
        return self.bev

    __call__ = forward

    def dense_codes(self, bev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 codes ``[agents, C * D, H, W]`` of the (unpadded) map -- what ``HeightCompression`` returns before dequantization."""
        b = self.bev if bev is None else bev
        return (b[:, 1:-1, 1:-1, :].to(torch.int16) + 128).to(torch.uint8).permute(0, 3, 1, 2).contiguous()

    def dequant(self, bev: Optional[torch.Tensor] = None) -> torch.Tensor:
        d, z = self.out_q
        return (self.dense_codes(bev).to(torch.float32) - z) * d


def deploy_second(qt_encoder=None, state=None, device="cuda:0", **kw) -> DeployedSecondEncoder:
    """``qt_encoder``: a calibrated ``QuantSECOND`` (or anything with ``.encoder_m1`` holding one)."""
    if state is None:
        from .ptq_state import export_second_state
        enc = getattr(getattr(qt_encoder, "model", qt_encoder), "encoder_m1", qt_encoder)
        state = export_second_state(enc)
    return DeployedSecondEncoder(state, device=device, **kw)
