"""GPU pre-step: LiDAR sweeps -> the ``inputs_m1`` dict of the model.

Replaces ``SpVoxelPreprocessor.preprocess`` + ``collate_batch`` (``pre_processor/sp_voxel_preprocessor.py:54-85,109-174``)
for the deployed path: the points stay on the device, ``qv2x_voxelize_f32`` builds the padded pillars.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import lib as L


class GpuVoxelizer:
    def __init__(self, lidar_range: Sequence[float], voxel_size: Sequence[float], max_points: int = 32,
                 max_voxels: int = 70000, device="cuda"):
        self.lib = L.load()
        self.range = (C.c_float * 6)(*[float(v) for v in lidar_range])
        self.vsize = (C.c_float * 3)(*[float(v) for v in voxel_size])
        self.max_points, self.max_voxels = int(max_points), int(max_voxels)
        self.dev = torch.device(device)
        self._ws = None

    def one(self, points: torch.Tensor, agent: int):
        """points f32 [P, 4] on the device -> (voxel_features [M,32,4], voxel_coords [M,4] (agent,z,y,x), voxel_num_points [M])"""
        points = points.contiguous()
        p = int(points.shape[0])
        if p == 0:                       # an empty sweep: no pillar (the contract oracle returns empty arrays too)
            return (torch.empty((0, self.max_points, 4), dtype=torch.float32, device=self.dev),
                    torch.empty((0, 4), dtype=torch.int32, device=self.dev), torch.empty((0,), dtype=torch.int32, device=self.dev))
        need = int(self.lib.qv2x_voxelize_workspace_bytes(p))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        feats = torch.empty((self.max_voxels, self.max_points, 4), dtype=torch.float32, device=self.dev)
        coords = torch.empty((self.max_voxels, 4), dtype=torch.int32, device=self.dev)
        nump = torch.empty((self.max_voxels,), dtype=torch.int32, device=self.dev)
        count = torch.zeros(1, dtype=torch.int32, device=self.dev)
        L.check(self.lib.qv2x_voxelize_f32(L.ptr(points), p, self.range, self.vsize, agent, self.max_points, self.max_voxels,
                                           L.ptr(self._ws), need, L.ptr(feats), L.ptr(coords), L.ptr(nump), L.ptr(count),
                                           L.current_stream()), "qv2x_voxelize_f32")
        m = int(count.item())            # the one host sync of the pre-step: the pillar count sizes every later launch
        return feats[:m], coords[:m], nump[:m]

    def fixed(self, sweeps: Sequence[torch.Tensor], capacity: int) -> dict:
        """HIP-graph capturable form: one sweep per batch index, EVERY sweep's ``capacity`` rows handed on (rows past a sweep's voxel count
        carry batch index -1, which the PFN kernel drops), no host read-back.  The returned dict also holds ``voxel_counts`` i32 [n] on
        the device.  Sweeps must keep their point counts between replays of a captured graph."""
        n = len(sweeps)
        feats = torch.empty((n * capacity, self.max_points, 4), dtype=torch.float32, device=self.dev)
        coords = torch.empty((n * capacity, 4), dtype=torch.int32, device=self.dev)
        nump = torch.empty((n * capacity,), dtype=torch.int32, device=self.dev)
        counts = torch.zeros(n, dtype=torch.int32, device=self.dev)
        need = max(int(self.lib.qv2x_voxelize_workspace_bytes(int(p.shape[0]))) for p in sweeps)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        for a, pts in enumerate(sweeps):
            pts = pts.contiguous()
            L.check(self.lib.qv2x_voxelize_f32(L.ptr(pts), int(pts.shape[0]), self.range, self.vsize, a, self.max_points, capacity,
                                               L.ptr(self._ws), need, L.ptr(feats[a * capacity:]), L.ptr(coords[a * capacity:]),
                                               L.ptr(nump[a * capacity:]), L.ptr(counts[a:]), L.current_stream()), "qv2x_voxelize_f32")
        return {"voxel_features": feats, "voxel_coords": coords, "voxel_num_points": nump, "voxel_counts": counts}

    def __call__(self, sweeps: Sequence[torch.Tensor]) -> dict:
        """One sweep per agent -> ``inputs_m1``; agents are the batch index, as ``collate_batch`` does."""
        parts = [self.one(pts, a) for a, pts in enumerate(sweeps)]
        return {"voxel_features": torch.cat([p[0] for p in parts]),
                "voxel_coords": torch.cat([p[1] for p in parts]),
                "voxel_num_points": torch.cat([p[2] for p in parts])}
