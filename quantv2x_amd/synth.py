"""Seeded synthetic inputs and weights for the QuantV2X hot path.

There is no dataset and no checkpoint in this environment, so every test, the
golden-vector generator (``tests/golden/make_golden.py``) and ``bench.py`` build
their inputs here, from ``numpy.random.Generator(PCG64(seed))`` only (torch RNG is
never used, so the reference and this build construct identical tensors from a
seed).

What is mirrored from the reference (shapes/keys only, no code):
  * ``model.args`` schema of ``hypes_yaml/v2x_real/Codebook/Attfuse/lidar_attfuse_stage3.yaml:110-157``
    and ``hypes_yaml/opv2v/LiDAROnly/lidar_attfuse.yaml`` (ranges/voxel sizes).
  * the ``data_dict`` keys produced by ``intermediate_heter_fusion_dataset.collate_batch_test``
    (``inputs_m1{voxel_features,voxel_coords,voxel_num_points}``, ``agent_modality_list``,
    ``record_len``, ``pairwise_t_matrix``), SURVEY.md §8(b).
  * the voxelizer *contract* of ``spconv.utils.Point2VoxelCPU3d`` as used by
    ``pre_processor/sp_voxel_preprocessor.py:54-85``: voxels in order of first
    point, at most ``max_points`` points per voxel (first come), at most
    ``max_voxels`` voxels, coords ``(z, y, x)``, zero padded.  spconv itself is
    not vendored by the reference, so this is a contract, not a parity claim.
"""
from __future__ import annotations

import copy
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- shapes

SHAPES = {
    # name: (lidar_range, voxel_size, max_voxels, max_cav)
    "v2xreal": ([-140.8, -40.0, -3.0, 140.8, 40.0, 1.0], [0.4, 0.4, 4.0], 70000, 5),
    "opv2v": ([-102.4, -102.4, -3.0, 102.4, 102.4, 1.0], [0.4, 0.4, 4.0], 70000, 8),
    # 64 x 32 voxel grid -> 32 x 16 feature map; used for golden vectors / CPU tests
    "tiny": ([-12.8, -6.4, -3.0, 12.8, 6.4, 1.0], [0.4, 0.4, 4.0], 2048, 5),
    # 128 x 64 grid -> 64 x 32 map; mid-size parity case
    "small": ([-25.6, -12.8, -3.0, 25.6, 12.8, 1.0], [0.4, 0.4, 4.0], 8192, 5),
}


# SECOND encoder: 0.1 m voxels, 40 height slices, <= 5 points per voxel (the HEAL / OPV2V SECOND yaml)
SECOND_SHAPES = {
    # name: (lidar_range, voxel_size, max_voxels)
    "second_tiny": ([-12.8, -6.4, -3.0, 12.8, 6.4, 1.0], [0.1, 0.1, 0.1], 4096),        # 256 x 128 x 40 -> 32 x 16 BEV map (= "tiny")
    "second_small": ([-25.6, -12.8, -3.0, 25.6, 12.8, 1.0], [0.1, 0.1, 0.1], 16384),    # 512 x 256 x 40 -> 64 x 32
    "second_full": ([-140.8, -40.0, -3.0, 140.8, 40.0, 1.0], [0.1, 0.1, 0.1], 70000),   # 2816 x 800 x 40 -> 352 x 100
    "second_opv2v": ([-102.4, -102.4, -3.0, 102.4, 102.4, 1.0], [0.1, 0.1, 0.1], 70000),  # 2048 x 2048 x 40 -> 256 x 256 (= "opv2v")
}
SECOND_OF = {"tiny": "second_tiny", "small": "second_small", "v2xreal": "second_full", "opv2v": "second_opv2v"}


def make_second_args(shape: str = "second_tiny", num_features_out: int = 128) -> dict:
    """``encoder_args`` of a ``core_method: second`` modality (reference ``heter_encoders.py:52-64``)."""
    lidar_range, voxel_size, _ = SECOND_SHAPES[shape]
    return {"voxel_size": list(voxel_size), "lidar_range": list(lidar_range),
            "mean_vfe": {"num_point_features": 4},
            "spconv": {"num_features_in": 4, "num_features_out": int(num_features_out)},
            "map2bev": {"feature_num": 2 * int(num_features_out)}}


def make_second_scene(shape: str = "second_tiny", n_agents: int = 1, seed: int = 0, n_points: int = 4000) -> dict:
    """``inputs_m1`` of a SECOND modality: <= 5 points per 0.1 m voxel, ``voxel_coords`` = (agent, z, y, x)."""
    lidar_range, voxel_size, max_voxels = SECOND_SHAPES[shape]
    feats, coords, nump = [], [], []
    sigma = 0.25 * (lidar_range[3] - lidar_range[0])
    for a in range(n_agents):
        pts = make_points(lidar_range, n_points, seed * 131 + a, sigma_m=sigma)
        # a ground sheet and a few walls: neighbouring voxels must be occupied or every window holds a single site
        g = np.random.Generator(np.random.PCG64([seed, a, 7]))
        pts[: len(pts) // 2, 2] = lidar_range[2] + 0.35 + 0.15 * g.random(len(pts) // 2).astype(np.float32)
        f, c, n = voxelize(pts, lidar_range, voxel_size, max_points=5, max_voxels=max_voxels)
        feats.append(f)
        coords.append(np.concatenate([np.full((len(c), 1), a, np.int32), c], axis=1))
        nump.append(n)
    return {"voxel_features": np.concatenate(feats), "voxel_coords": np.concatenate(coords), "voxel_num_points": np.concatenate(nump)}


def grid_size(lidar_range: Sequence[float], voxel_size: Sequence[float]) -> Tuple[int, int, int]:
    r = np.asarray(lidar_range, dtype=np.float64)
    g = np.round((r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float64)).astype(np.int64)
    return int(g[0]), int(g[1]), int(g[2])


def make_hypes(shape: str = "v2xreal", multiclass: bool = True, codebook: bool = True,
               supervise_single: bool = True, dict_size: int = 128, seg_num: int = 1, fusion: str = "att", compress_ratio: int = 0,
               encoder: str = "point_pillar", modalities: Sequence[str] = ("m1",), encoders: Optional[Dict[str, str]] = None) -> dict:
    """Return a ``hypes`` dict with the ``model`` section the reference's yaml would give.  ``modalities=("m1", "m2")``: a heterogeneous
    model (heter_model_baseline.py:41-75: one encoder / backbone / shrinker per modality; here both LiDAR PointPillar, same architecture,
    their own weights).  ``encoder="second"``: the m1 modality is the
    SECOND encoder over ``SECOND_SHAPES["second_" + shape]`` (same metric range, 0.1 m voxels); its 256-channel map is already at the
    resolution PointPillar's first backbone level reaches, so the backbone starts with stride 1 and ``inplanes: 256``.
    ``encoders={"m3": "second"}``: the encoder per modality (heter_model_baseline.py:47-59 picks it by each modality's ``core_method``) --
    a MIXED-encoder model, e.g. ``modalities=("m1", "m3")`` with PointPillar agents and SECOND agents in one scene."""
    lidar_range, voxel_size, max_voxels, max_cav = SHAPES[shape]
    args = {
        "ego_modality": "m1",
        "num_class": 3 if multiclass else 1,
        "lidar_range": list(lidar_range),
        "supervise_single": bool(supervise_single),
        "in_head_single": 256,
        "m1": {
            "core_method": "point_pillar",
            "sensor_type": "lidar",
            "encoder_args": {
                "voxel_size": list(voxel_size),
                "lidar_range": list(lidar_range),
                "pillar_vfe": {"use_norm": True, "with_distance": False,
                               "use_absolute_xyz": True, "num_filters": [64]},
                "point_pillar_scatter": {"num_features": 64},
            },
            "backbone_args": {
                "layer_nums": [3, 5, 8], "layer_strides": [2, 2, 2],
                "num_filters": [64, 128, 256], "upsample_strides": [1, 2, 4],
                "num_upsample_filter": [128, 128, 128],
            },
            "shrink_header": {"kernal_size": [3], "stride": [1], "padding": [1],
                              "dim": [256], "input_dim": 384},
        },
        "fusion_method": fusion,              # "att" (AttFusion) | "max" (F-Cooper's MaxFusion, hypes_yaml/v2x_real/Codebook/Fcooper)
        "att": {"feat_dim": 256},
        "in_head": 256,
        "anchor_number": 2,
        "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]},
    }
    if codebook:
        args["codebook"] = {"seg_num": seg_num, "dict_size": dict_size}
        args["use_codebook"] = True
    if compress_ratio:                                # hypes_yaml/v2x_real/Naive_Compressor/*: `compressor: {input_dim: 256, compress_ratio: 16}`
        args["compressor"] = {"input_dim": 256, "compress_ratio": int(compress_ratio)}
    if encoder not in ("second", "point_pillar"):
        raise ValueError(encoder)
    kinds = {m: encoder for m in modalities}
    kinds.update(encoders or {})
    pillar_cfg = copy.deepcopy(args["m1"])
    for m in modalities:
        args[m] = copy.deepcopy(pillar_cfg)
        if kinds[m] == "second":
            args[m]["core_method"] = "second"
            args[m]["encoder_args"] = make_second_args(SECOND_OF[shape])
            args[m]["backbone_args"].update({"layer_strides": [1, 2, 2], "inplanes": 256})
        elif kinds[m] != "point_pillar":
            raise ValueError(kinds[m])
    if "m1" not in modalities:
        del args["m1"]
        args["ego_modality"] = modalities[0]
    core = "heter_baseline_collab_codebook" if codebook else "heter_model_baseline"
    if multiclass:
        core += "_mc"
    return {
        "name": f"synthetic_{shape}",
        "model": {"core_method": core, "args": args},
        "preprocess": {"args": {"voxel_size": list(voxel_size), "max_points_per_voxel": 32,
                                "max_voxel_test": max_voxels},
                       "cav_lidar_range": list(lidar_range)},
        "train_params": {"max_cav": max_cav},
    }


def make_pyramid_hypes(shape: str = "v2xreal", codebook: bool = True, dict_size: int = 128, seg_num: int = 1,
                       encdec: bool = True, multiclass: bool = True, modalities: Sequence[str] = ("m1",)) -> dict:
    """The ``model`` section of ``hypes_yaml/v2x_real/Codebook/Pyramid/lidar_pyramid_stage3.yaml:110-157`` (HEAL Pyramid fusion,
    ResNeXt levels, 64-channel codebook) at one of ``SHAPES``; ``hard_eval`` as in ``make_hypes``' codebook models.
    ``multiclass=False``: the single-class ``heter_pyramid_collab[_codebook]`` of the OPV2V / DAIR-V2X yamls (no ``num_class`` argument,
    2 | 14 | 4 head channels).  ``modalities=("m1", "m2")``: HEAL's heterogeneous setting -- one encoder / ResNet backbone / aligner per
    modality (heter_pyramid_collab_mc.py:45-86), here both LiDAR PointPillar with their own weights."""
    lidar_range, voxel_size, max_voxels, max_cav = SHAPES[shape]
    args = {
        "num_class": 3,
        "lidar_range": list(lidar_range),
        "supervise_single": True,
        "m1": {
            "core_method": "point_pillar",
            "sensor_type": "lidar",
            "encoder_args": {
                "voxel_size": list(voxel_size),
                "lidar_range": list(lidar_range),
                "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
                "point_pillar_scatter": {"num_features": 64},
            },
            "backbone_args": {"layer_nums": [3], "layer_strides": [2], "num_filters": [64]},
            "aligner_args": {"core_method": "identity"},
        },
        "fusion_backbone": {
            "resnext": True, "stage": "collab", "layer_nums": [3, 5, 8], "layer_strides": [1, 2, 2],
            "num_filters": [64, 128, 256], "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128],
            "anchor_number": 2,
        },
        "shrink_header": {"kernal_size": [3], "stride": [1], "padding": [1], "dim": [256], "input_dim": 384},
        "fusion_method": "pyramid",
        "in_head": 256,
        "anchor_number": 2,
        "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]},
    }
    m1 = args["m1"]
    for m in modalities:
        args[m] = copy.deepcopy(m1)
    if "m1" not in modalities:
        del args["m1"]
    if not multiclass:
        del args["num_class"]
    core = "heter_pyramid_collab_mc" if multiclass else "heter_pyramid_collab"
    if codebook:
        args["codebook"] = {"seg_num": seg_num, "dict_size": dict_size, "hard_eval": True}
        args["use_codebook"] = True
        core = ("heter_pyramid_collab_codebook_mc" + ("_encdec" if encdec else "")) if multiclass else "heter_pyramid_collab_codebook"
    return {
        "name": f"synthetic_pyramid_{shape}",
        "model": {"core_method": core, "args": args},
        "preprocess": {"args": {"voxel_size": list(voxel_size), "max_points_per_voxel": 32, "max_voxel_test": max_voxels},
                       "cav_lidar_range": list(lidar_range)},
        "train_params": {"max_cav": max_cav},
    }


# --------------------------------------------------------------------------- weights

# The V2X-Real multi-class post-processor configuration (hypes_yaml/v2x_real/*: three anchor sets, VoxelPostprocessor3Heads)
MC_ANCHOR_CFGS = [dict(class_name=n, anchor_sizes=[sz], anchor_rotations=[0, 1.57], anchor_bottom_heights=[zb], align_center=True,
                       feature_map_stride=2, matched_threshold=0.6, unmatched_threshold=0.45)
                  for n, sz, zb in (("vehicle", [3.9, 1.6, 1.56], -1.78), ("pedestrian", [0.8, 0.6, 1.73], -0.6), ("truck", [8, 3, 3], -1.78))]


def mc_postprocess_params(lidar_range: Sequence[float], grid_w: int, grid_h: int) -> dict:
    return {"core_method": "VoxelPostprocessor3Heads", "gt_range": list(lidar_range), "order": "hwl", "nms_thresh": 0.15,
            "anchor_args": {"cav_lidar_range": list(lidar_range), "W": grid_w, "H": grid_h, "anchor_generator_config": MC_ANCHOR_CFGS},
            "target_args": {"score_threshold": 0.2}}


def _rng(seed: int, tag: str) -> np.random.Generator:
    # independent stream per parameter name so that key order never matters
    h = np.frombuffer(tag.encode(), dtype=np.uint8).astype(np.uint64)
    mix = int((h * (np.arange(h.size, dtype=np.uint64) + np.uint64(131))).sum() % np.uint64(2**31 - 1))
    return np.random.Generator(np.random.PCG64([seed, mix]))


def make_state_dict(template: Dict[str, "object"], seed: int = 0) -> Dict[str, np.ndarray]:
    """Generate values for every key of ``template`` (a ``state_dict()`` of the
    reference model or of this build's plugin model -- the key sets are equal).

    He-normal convolutions / linears, BatchNorm gamma in [0.8, 1.2], small beta,
    non-trivial running statistics, so that every activation quantizer sees a
    healthy range (SURVEY.md §7 "No checkpoint": torch's default init collapses
    the activation ranges to the 1e-8 floor by block 2).
    """
    out: Dict[str, np.ndarray] = {}
    for key in sorted(template.keys()):
        shape = tuple(int(s) for s in template[key].shape)
        g = _rng(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.asarray(100, dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = g.normal(0.0, 0.1, shape).astype(np.float32)
        elif leaf == "running_var":
            out[key] = g.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "_codebook":  # [m, k, d]; SmallInit as codebook.py:312-313 but wider so codes spread
            d_total = shape[0] * shape[2]
            out[key] = g.normal(0.0, 2.0 * math.sqrt(2.0 / (5.0 * d_total)), shape).astype(np.float32)
        elif leaf == "_temperature":
            out[key] = np.ones(shape, dtype=np.float32)
        elif key.startswith("codebook._freqEMA") or "._freqEMA." in key:
            out[key] = np.full(shape, 1.0 / shape[-1], dtype=np.float32)
        elif leaf == "weight" and len(shape) == 1 and ".resnet." in key and (
                ".bn3." in key or (".bn2." in key and key.startswith("backbone_m"))):
            out[key] = g.uniform(0.2, 0.4, shape).astype(np.float32)   # last BN of a residual branch: keeps 16 stacked blocks O(1)
        elif leaf == "weight" and len(shape) == 1:  # BatchNorm gamma
            out[key] = g.uniform(0.8, 1.2, shape).astype(np.float32)
        elif leaf == "bias":
            out[key] = g.normal(0.0, 0.05, shape).astype(np.float32)
        elif leaf == "weight":
            if ".deblocks." in key and len(shape) == 4:  # ConvTranspose2d [Cin, Cout, k, k]: one tap per output
                fan_in = shape[0]
            elif len(shape) == 5:                        # sparse 3-D convolution [Cout, kz, ky, kx, Cin]; about a third of a window is occupied
                fan_in = shape[1] * shape[2] * shape[3] * shape[4] // 3
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
            else:
                fan_in = shape[1]
            std = math.sqrt(2.0 / fan_in)
            if key.startswith("codebook."):  # plain linear heads, no ReLU: keep variance ~1
                std = math.sqrt(1.0 / fan_in)
            out[key] = g.normal(0.0, std, shape).astype(np.float32)
        else:  # buffers we do not know: leave numerically harmless
            out[key] = np.zeros(shape, dtype=np.float32)
    return out


def make_state_dict_contractive(template: Dict[str, "object"], seed: int = 0, pass_gain: float = 0.9, noise: float = 0.25) -> Dict[str, np.ndarray]:
    """A weight set on which a +-1 code flip is NOT amplified by the convolution stack (tests/test_hip_box_agreement.py's yardstick).

    With He-normal weights one flipped input code shifts each of its 9 x C_out fan-out sums by ~N(0, 1/sqrt(9 C_in)) output LSBs, i.e.
    ~0.8 sqrt(18 C_out^2 / C_in) ~ 27 expected new flips per flip and layer: the integer path and the reference's fp32-emulated path
    decorrelate until ~5 % of the codes differ, whatever the implementation.  Here every backbone / shrinker convolution is
    ``pass_gain`` x (centre tap, channel co <- co mod C_in) plus a random part whose fan-out L1 norm is ``noise``: one flip in makes
    ~pass_gain + noise flips out, so differences stay at the per-layer floor of fresh flips.  BatchNorms are the identity, deconvolutions
    copy channel ci to co = ci mod C_out on every tap.  PFN, codebook and heads keep ``make_state_dict``'s values."""
    out = make_state_dict(template, seed)
    for key in sorted(template.keys()):
        if not key.startswith(("backbone_m", "shrinker_m")):
            continue
        shape = tuple(int(v) for v in template[key].shape)
        leaf = key.rsplit(".", 1)[-1]
        g = _rng(seed + 7919, key)
        if leaf == "running_mean":
            out[key] = np.zeros(shape, np.float32)
        elif leaf == "running_var":
            out[key] = np.ones(shape, np.float32)          # (eps = 1e-3 makes the fold 0.9995: harmless)
        elif leaf == "weight" and len(shape) == 1:
            out[key] = np.ones(shape, np.float32)
        elif leaf == "bias":
            out[key] = np.full(shape, 0.02, np.float32)
        elif leaf == "weight" and len(shape) == 4:
            deconv = ".deblocks." in key
            fan_out = (shape[1] if deconv else shape[0] * shape[2] * shape[3])      # outputs one input value reaches
            w = g.normal(0.0, noise / (0.8 * fan_out), shape).astype(np.float32)
            if deconv:                                     # [C_in, C_out, k, k]: every tap copies ci -> ci mod C_out
                ci = np.arange(shape[0])
                w[ci, ci % shape[1], :, :] += pass_gain
            else:                                          # [C_out, C_in, 3, 3]
                co = np.arange(shape[0])
                w[co, co % shape[1], shape[2] // 2, shape[3] // 2] += pass_gain
            out[key] = w
    return out


def load_state_dict_numpy(model, sd: Dict[str, np.ndarray]) -> None:
    import torch
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"state_dict mismatch: missing={missing} unexpected={unexpected}")


# --------------------------------------------------------------------------- points / pillars

def make_points(lidar_range: Sequence[float], n_points: int, seed: int,
                sigma_m: float = 35.0) -> np.ndarray:
    """Synthetic LiDAR sweep: r ~ |N(0, sigma)|, theta ~ U(0, 2pi),
    z ~ U(zmin, zmax), intensity ~ U(0, 1); cropped to ``lidar_range``.  float32 [n, 4]."""
    g = np.random.Generator(np.random.PCG64([seed, 0xC0FFEE]))
    r = np.abs(g.normal(0.0, sigma_m, n_points))
    th = g.uniform(0.0, 2.0 * np.pi, n_points)
    x = r * np.cos(th)
    y = r * np.sin(th)
    z = g.uniform(lidar_range[2], lidar_range[5], n_points)
    inten = g.uniform(0.0, 1.0, n_points)
    pts = np.stack([x, y, z, inten], axis=1).astype(np.float32)
    lo = np.asarray(lidar_range[:3], dtype=np.float32)
    hi = np.asarray(lidar_range[3:], dtype=np.float32)
    keep = np.all((pts[:, :3] > lo) & (pts[:, :3] < hi), axis=1)
    return np.ascontiguousarray(pts[keep])


def voxelize(points: np.ndarray, lidar_range: Sequence[float], voxel_size: Sequence[float],
             max_points: int = 32, max_voxels: int = 70000):
    """Dense-grid pillar voxelizer following the Point2VoxelCPU3d contract (see module doc).

    Returns ``voxel_features f32 [M, max_points, 4]``, ``voxel_coords i32 [M, 3] (z, y, x)``,
    ``voxel_num_points i32 [M]``.
    """
    nx, ny, nz = grid_size(lidar_range, voxel_size)
    lo = np.asarray(lidar_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    c = np.floor((points[:, :3] - lo) / vs).astype(np.int64)
    ok = (c[:, 0] >= 0) & (c[:, 0] < nx) & (c[:, 1] >= 0) & (c[:, 1] < ny) & (c[:, 2] >= 0) & (c[:, 2] < nz)
    idx_pts = np.nonzero(ok)[0]
    c = c[ok]
    lin = (c[:, 2] * ny + c[:, 1]) * nx + c[:, 0]
    # voxel order = order of first appearance
    uniq, first, inv = np.unique(lin, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # uniq index -> rank by first appearance
    rank_of_uniq = np.empty_like(order)
    rank_of_uniq[order] = np.arange(order.size)
    vox_id = rank_of_uniq[inv]                        # per point: voxel number
    keep_vox = vox_id < max_voxels
    idx_pts, vox_id = idx_pts[keep_vox], vox_id[keep_vox]
    # slot of each point inside its voxel = number of earlier points in the same voxel
    srt = np.argsort(vox_id, kind="stable")
    vsort = vox_id[srt]
    start = np.r_[0, np.nonzero(np.diff(vsort))[0] + 1]
    counts = np.diff(np.r_[start, vsort.size])
    slot_sorted = np.arange(vsort.size) - np.repeat(start, counts)
    slot = np.empty_like(slot_sorted)
    slot[srt] = slot_sorted
    sel = slot < max_points
    m = int(min(order.size, max_voxels))
    feats = np.zeros((m, max_points, points.shape[1]), dtype=np.float32)
    feats[vox_id[sel], slot[sel]] = points[idx_pts[sel]]
    nump = np.minimum(np.bincount(vox_id, minlength=m), max_points).astype(np.int32)
    lin_first = uniq[order][:m]
    cz = lin_first // (ny * nx)
    cy = (lin_first // nx) % ny
    cx = lin_first % nx
    coords = np.stack([cz, cy, cx], axis=1).astype(np.int32)
    return feats, coords, nump


# --------------------------------------------------------------------------- poses

def pose_matrix(x: float, y: float, yaw: float) -> np.ndarray:
    c, s = math.cos(yaw), math.sin(yaw)
    t = np.eye(4, dtype=np.float64)
    t[0, 0], t[0, 1], t[1, 0], t[1, 1] = c, -s, s, c
    t[0, 3], t[1, 3] = x, y
    return t


def agent_poses(n_agents: int, layout: str = "line") -> List[np.ndarray]:
    """World poses of the agents.  'line': agent j at yaw 0.1*j, t=(5j, -2j) m;
    'ring': 30 m ring with mixed yaw (config 4)."""
    poses = []
    for j in range(n_agents):
        if layout == "ring" and j > 0:
            a = 2.0 * math.pi * j / max(n_agents, 2)
            poses.append(pose_matrix(30.0 * math.cos(a) - 30.0, 30.0 * math.sin(a), 0.35 * j - 0.5))
        else:
            poses.append(pose_matrix(5.0 * j, -2.0 * j, 0.1 * j))
    return poses


def pairwise_t_matrix(poses: Sequence[np.ndarray], max_cav: int) -> np.ndarray:
    """``T[i, j] = T_j^-1 T_i`` identity padded to ``[L, L, 4, 4]`` (the layout
    ``utils/transformation_utils.py:21-66 get_pairwise_transformation`` produces)."""
    L = max_cav
    t = np.tile(np.eye(4, dtype=np.float64), (L, L, 1, 1))
    for i in range(len(poses)):
        for j in range(len(poses)):
            if i != j:
                t[i, j] = np.linalg.solve(poses[j], poses[i])
    return t


# --------------------------------------------------------------------------- scenes

def make_scene(shape: str = "v2xreal", n_agents: int = 1, seed: int = 0, n_points: int = 60000,
               layout: str = "line", sigma_m: Optional[float] = None, max_cav: Optional[int] = None, encoder: str = "point_pillar",
               modalities: Optional[Sequence[str]] = None, encoders: Optional[Dict[str, str]] = None) -> dict:
    """Build the numpy form of ``batch_data['ego']`` for one frame (batch size 1).  ``encoder="second"``: ``inputs_m1`` holds the 0.1 m
    voxels of ``make_second_scene`` over the same range.  ``modalities``: one name per agent (default all ``m1``); every modality gets its
    own ``inputs_<m>`` whose ``voxel_coords[:, 0]`` counts that modality's agents (what the reference's collate gives its encoders)."""
    lidar_range, voxel_size, max_voxels, L = SHAPES[shape]
    if max_cav is not None:
        L = max_cav
    L = max(L, n_agents)
    if sigma_m is None:
        sigma_m = 0.25 * (lidar_range[3] - lidar_range[0]) / 2.0 + 0.0
        sigma_m = {"v2xreal": 35.0, "opv2v": 45.0}.get(shape, sigma_m)
    modalities = ["m1"] * n_agents if modalities is None else list(modalities)
    assert len(modalities) == n_agents
    per = {m: ([], [], []) for m in dict.fromkeys(modalities)}
    for a in range(n_agents):
        pts = make_points(lidar_range, n_points, seed * 1000 + a, sigma_m)
        f, c, n = voxelize(pts, lidar_range, voxel_size, 32, max_voxels)
        feats, coords, nums = per[modalities[a]]
        coords.append(np.concatenate([np.full((c.shape[0], 1), len(feats), dtype=np.int32), c], axis=1))     # index among ITS modality's agents
        feats.append(f)
        nums.append(n)
    poses = agent_poses(n_agents, layout)
    out = {}
    for m, (feats, coords, nums) in per.items():
        out["inputs_" + m] = {"voxel_features": np.concatenate(feats, axis=0), "voxel_coords": np.concatenate(coords, axis=0),
                              "voxel_num_points": np.concatenate(nums, axis=0)}
    if encoder == "second":
        out["inputs_m1"] = make_second_scene(SECOND_OF[shape], n_agents, seed, n_points)
    for m, kind in (encoders or {}).items():                          # a mixed scene: this modality's agents as 0.1 m voxels of their own sweeps
        if kind == "second" and m in per:
            out["inputs_" + m] = make_second_scene(SECOND_OF[shape], len(per[m][0]), seed * 7 + 1, n_points)
    out.update({
        "agent_modality_list": modalities,
        "record_len": np.asarray([n_agents], dtype=np.int64),
        "pairwise_t_matrix": pairwise_t_matrix(poses, L)[None].astype(np.float64),
    })
    return out


def scene_to_torch(scene: dict, device="cpu") -> dict:
    import torch

    def conv(v):
        if isinstance(v, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(v)).to(device)
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        return copy.copy(v)

    return {k: conv(v) for k, v in scene.items()}
