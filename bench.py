#!/usr/bin/env python
"""bench.py -- frames/s of the quantized per-agent encode + intermediate-fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--inflight F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the whole hot path (a1-a11: PFN + scatter, int8 backbone + shrinker, codebook encode, exchange,
decode + warp + attention, heads) over one batch of B synthetic V2X-Real-shaped frames (default B = 32: the reference's
model contract has a batch dimension, ``record_len`` / ``pairwise_t_matrix[B]``), replayed as HIP graphs; at N = 1, F = 2
such batches are in flight on two streams (the next batch's PFN / backbone fill the tail of the previous batch's encode).
``value`` = frames per second over exactly K steps; the p50 latency of ONE frame run alone is reported next to it, and
``value_one_frame_at_a_time`` is its reciprocal throughput (round 1's ``value``).

With N GPUs, rank r owns agent r (one process per GPU): every step each rank encodes B frames of its agent, ONE RCCL
all-gather moves the code planes + poses, and every rank fuses as the ego of its own view, so a step yields N x B fused
N-agent frames; ``value`` = N * B * K / (max-over-ranks time).  Per-GPU work is fixed as N grows: ``"scaling": "weak"``.
``--rehearse-world W`` (one GPU): rank 0's step of the W-GPU line with the all-gather replaced by a device copy -- its own line, not a
multi-GPU measurement.

Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task prompt) with
  roofline         the dominant kernel by time (codebook_encode_wave_kernel: f32 MFMA), launch duration from HIP events
  roofline_stages  every stage of the frame against its own bound (int8 MFMA / f32 MFMA / HBM), same method
  cpu_baseline     the CPU oracle (``oracle/``, the checker -- never the product) on the host cores; and
  cpu_baseline_torch  the torch restatement of the reference (plugin mirror) in fp32 and W8A8 fake-quant, all host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: dense bf16 MFMA 2.5 PF (int8 = 2 x), f32 MFMA = f32 vector peak, HBM3E 8 TB/s
INT8_MFMA_PEAK_TOPS = 5000.0
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
SHAPE = "v2xreal"                          # N = 1 default; main() re-binds SHAPE / N_POINTS from workload_for(world)
N_POINTS = 60000
ENCODE_GFLOP_PER_AGENT_FRAME = 43.84       # 21.92 GMAC fp32, reference op order (V2X-Real: 35 200 cells)
ENCODE_GMAC_PER_CELL = 21.92e9 / 35200     # the same figure per feature-map cell (OPV2V: 65 536 cells -> 40.8 GMAC)


def workload_for(world):
    """BASELINE.json ``configs`` -> the concrete synthetic workload of an N-GPU run:
    N = 1 configs[1]; N = 2 configs[2] (V2X-Real, line layout); N = 3..4 configs[3] (V2X-Real VC, ring layout); both with
    max_cav 5 and the multi-class (mc) heads of lidar_attfuse_stage3.yaml:12.  N >= 5 configs[4]: the OPV2V(-H) 512 x 512 grid
    (opv2v/LiDAROnly/lidar_attfuse.yaml:17), max_cav raised to 8, single-class heads, sweeps dense enough for >= 40k pillars."""
    if world == 1:
        return {"index": 1, "shape": "v2xreal", "layout": "line", "n_points": 60000, "multiclass": True, "max_cav": 5,
                "workload": "Single-agent int8 PointPillar + BEV backbone on 1xMI355X, synthetic V2X-Real point cloud (~60k pts, 0.4 m voxels)",
                "grid": "704x200x1 voxels -> 256x100x352 feature map"}
    if world == 2:
        return {"index": 2, "shape": "v2xreal", "layout": "line", "n_points": 60000, "multiclass": True, "max_cav": 5,
                "workload": "2-agent intermediate fusion with codebook-compressed BEV features, 2xMI355X, RCCL all-gather over xGMI "
                            "(V2X-Real shape, line layout, max_cav 5, mc heads)",
                "grid": "704x200x1 voxels -> 256x100x352 feature map"}
    if world <= 4:
        return {"index": 3, "shape": "v2xreal", "layout": "ring", "n_points": 60000, "multiclass": True, "max_cav": 5,
                "workload": f"{world}-agent V2X-Real VC scenario, int8 attentive fusion + spatial transform, {world}xMI355X "
                            "(ring layout, max_cav 5, mc heads)",
                "grid": "704x200x1 voxels -> 256x100x352 feature map"}
    return {"index": 4, "shape": "opv2v", "layout": "ring", "n_points": 70000, "multiclass": False, "max_cav": 8,
            "workload": f"{world}-agent OPV2V-H dense scene, full int8 pipeline with codebook compressor, {world}xMI355X "
                        "(512x512 grid, max_cav 8, single-class heads, >= 40k pillars per agent)",
            "grid": "512x512x1 voxels -> 256x256x256 feature map"}


def build_engine(n_threads, multiclass=True):
    import copy
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(n_threads)
    # the reference's flow: create_model -> load weights (seeded synthetic: no checkpoint exists here) -> QuantModel
    # -> weight quantizers -> one min-max observer pass (torch, on the host) -> freeze -> deploy on the HIP path
    model = train_utils.create_model(copy.deepcopy(synth.make_hypes(SHAPE, multiclass=multiclass))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    fp_model = copy.deepcopy(model)
    calib = synth.scene_to_torch(synth.make_scene(SHAPE, n_agents=1, seed=3, n_points=N_POINTS))
    qt = inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib])
    state = export_ptq_state(qt)
    return state, deploy(state=state), fp_model, qt


def frame_batch(world, rank, frames, device, layout=None, max_cav=None, own_only=False):
    """`frames` scenes of `world` agents (different sweeps, same poses).  Returns the numpy scene 0, the model input of a
    single-GPU batch (every agent of every frame, batch index = frame * world + agent), this rank's input (its own agent of every
    frame, batch index = frame) and the agents' world poses.  ``own_only`` (the ranks of an N-GPU run): only this rank's agent is
    generated -- the same sweeps ``synth.make_scene`` gives it (seed = scene seed * 1000 + agent) -- and the first two results are None."""
    import numpy as np
    import torch
    from quantv2x_amd import synth
    layout = layout or ("ring" if world > 2 else "line")
    cat = lambda parts: {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).to(device) for k in parts[0]}
    if own_only:
        lidar_range, voxel_size, max_voxels, _ = synth.SHAPES[SHAPE]
        sigma = {"v2xreal": 35.0, "opv2v": 45.0}[SHAPE]
        mine_parts = []
        for f in range(frames):
            vf, co, nump = synth.voxelize(synth.make_points(lidar_range, N_POINTS, (3 + f) * 1000 + rank, sigma), lidar_range, voxel_size, 32, max_voxels)
            mine_parts.append({"voxel_features": vf, "voxel_coords": np.concatenate([np.full((co.shape[0], 1), f, dtype=np.int32), co], axis=1),
                               "voxel_num_points": nump})
        return None, None, cat(mine_parts), synth.agent_poses(world, layout)
    scenes = [synth.make_scene(SHAPE, n_agents=world, seed=3 + f, n_points=N_POINTS, layout=layout, max_cav=max_cav) for f in range(frames)]
    full_parts, mine_parts = [], []
    for f, sc in enumerate(scenes):
        co = sc["inputs_m1"]["voxel_coords"]
        part = {k: v.copy() for k, v in sc["inputs_m1"].items()}
        part["voxel_coords"][:, 0] += f * world
        full_parts.append(part)
        sel = co[:, 0] == rank
        m = {k: v[sel].copy() for k, v in sc["inputs_m1"].items()}
        m["voxel_coords"][:, 0] = f
        mine_parts.append(m)
    full = {"inputs_m1": cat(full_parts), "agent_modality_list": ["m1"] * (world * frames),
            "record_len": torch.full((frames,), world, dtype=torch.int64),
            "pairwise_t_matrix": torch.from_numpy(np.concatenate([sc["pairwise_t_matrix"] for sc in scenes])).to(device)}
    return scenes[0], full, cat(mine_parts), synth.agent_poses(world, layout)


def event_time_us(fn, iters, launches_per_call=1):
    """average duration of one launch: HIP events on the launch stream around `iters` back-to-back calls"""
    import torch
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * launches_per_call)


def rooflines(eng, full, frames, iters):
    """``roofline`` (the dominant kernel) and ``roofline_stages``: live HIP-event time of every stage at the bench's batch
    size against the algorithmic work of SURVEY.md §8(d) x the agent-frames one launch processes."""
    import torch
    n = frames                                            # one agent per frame at N = 1
    hw = eng.fh * eng.fw
    inp = full["inputs_m1"]
    pillars = int(inp["voxel_features"].shape[0])
    plan = eng.conv_plan(n)
    bb = [p for p in plan if p[0] in ("conv", "chain") and p[1].name.startswith("backbone")]
    sh = [p for p in plan if p[0] == "conv" and p[1].name.startswith("shrinker")]
    de = [p for p in plan if p[0] == "deconv"]
    stages = {}

    def stage(name, fn, bound, work, unit, peak, launches, note):
        us = event_time_us(fn, iters)
        ach = work / (us * 1e-6) / (1e12 if unit != "GB/s" else 1e9)
        stages[name] = {"bound": bound, "us_per_batch": round(us, 2), "launches": launches, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 4), "work": note}

    def pfn_resident():                             # as the engine runs it: scatter into the resident clean canvas, un-scatter afterwards
        eng.pillars_to_canvas(inp, n, resident=True)
        eng.clear_pillars(inp, n)
    pfn_resident()
    stage("pfn_scatter+clear", pfn_resident, "hbm",
          pillars * (32 * 16 + 16 + 4) + 2 * pillars * 64, "GB/s", HBM_PEAK_GBS, 2,
          f"bytes actually moved: {pillars} pillars x 532 B read + their 64-byte cells written twice (scatter, then set back: the canvas stays "
          f"resident and clean).  SURVEY 8(d) a1+a2 counts {n} x 9.0 MB of canvas writes on top, which this design no longer performs")
    stage("backbone_convs_i8", lambda: eng.run_plan(n, only=lambda k, l: k in ("conv", "chain") and l.name.startswith("backbone")), "mfma-i8",
          sum(2.0 * p[7] for p in bb), "TOP/s", INT8_MFMA_PEAK_TOPS, len(bb),
          f"{sum(2.0 * p[7] for p in bb) / 1e9:.1f} GOP: 19 conv layers in {len(bb)} launches"
          + (" (level 0 = one fused launch)" if any(p[0] == "chain" for p in bb) else " (one per layer: the level-0 fusion only pays for a single agent-frame)"))
    stage("backbone_deconvs_f32", lambda: eng.run_plan(n, only=lambda k, l: k == "deconv"), "mfma-f32",
          sum(2.0 * p[7] for p in de), "TFLOP/s", F32_MFMA_PEAK_TFLOPS, 1, f"{sum(2.0 * p[7] for p in de) / 1e9:.2f} GFLOP: 3 deblocks, one launch")
    stage("shrinker_convs_i8", lambda: eng.run_plan(n, only=lambda k, l: k == "conv" and l.name.startswith("shrinker")), "mfma-i8",
          sum(2.0 * p[7] for p in sh), "TOP/s", INT8_MFMA_PEAK_TOPS, len(sh), f"{sum(2.0 * p[7] for p in sh) / 1e9:.1f} GOP: 3x3 384->256 + 3x3 256->256")
    enc_gflop = round(2.0 * ENCODE_GMAC_PER_CELL * hw / 1e9, 2)          # 43.84 at V2X-Real (35 200 cells), 81.61 at OPV2V (65 536)
    two_stage = eng.encode_mode == "two_stage" and eng.encode_form == "auto"
    if two_stage:
        # a6 as the engine runs it (round 6): exact integer candidates for every cell, the reference-order chain for the cells a proven bound
        # cannot decide -- the SAME indices (encode_two_stage.py).  Three entries: the pair of launches, and each stage by itself on the work it
        # EXECUTES (so that no fraction exceeds 1); the reference-order flops of the whole batch are stated beside them.
        import ctypes as C
        from quantv2x_amd import lib as L
        eng.encode_codes(n)
        torch.cuda.synchronize()
        ref = eng.encode_refine_stats(n)
        b = eng._workspace(n)
        gp, bias, tab, tau, _ = eng._two_stage
        d = L.EncodeDesc()
        d.n, d.h, d.w, d.levels, d.kc, d.segs = n, eng.fh, eng.fw, eng.enc_levels, eng.kc, 1
        d.in_zx, d.in_delta = int(eng.shrink1.out_q[1]), float(eng.shrink1.out_q[0])

        def stage1():
            L.check(eng.lib.qv2x_codebook_encode_candidates_i8(C.byref(d), L.ptr(b["s1"]), L.ptr(gp), L.ptr(bias), L.ptr(tab), tau, L.ptr(b["codes"]),
                                                               L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]), L.current_stream()), "candidates")

        def stage2():                                    # (the list of the last stage 1 stays on the device: the same cells every time)
            L.check(eng.lib.qv2x_codebook_encode_listed_f32(C.byref(d), L.ptr(b["s1"]), eng.level_ptrs, L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]),
                                                            L.ptr(b["codes"]), L.current_stream()), "listed")
        cand_ops = 2.0 * 3 * 256 * eng.enc_levels * eng.kc * n * hw
        stage("codebook_encode_candidates_i8", stage1, "mfma-i8", cand_ops, "TOP/s", INT8_MFMA_PEAK_TOPS, 1,
              f"stage 1: {n * hw} cells x (256 -> {eng.enc_levels * eng.kc} scores x 3 int8 limbs) = {cand_ops / 1e9:.1f} GOP on v_mfma_i32_32x32x32_i8, fp64 packed "
              f"argmin chain; lists the cells whose top-2 gap does not exceed the proven bound")
        stage1()
        # a cell first undecided at level c runs the latent chain of every level but the quantization head + distances only from level c on
        # (98 304 of a level's 229 376 MACs skipped per proven level; 622 592 per cell in all = ENCODE_GMAC_PER_CELL)
        listed_flop = 2.0 * ENCODE_GMAC_PER_CELL * sum(nc * (622592.0 - 98304.0 * c) / 622592.0 for c, nc in enumerate(ref["first_flagged_at_level"]))
        stage("codebook_encode_listed_f32", stage2, "mfma-f32", listed_flop, "TFLOP/s", F32_MFMA_PEAK_TFLOPS, 1,
              f"stage 2: the {ref['refined']} listed cells ({ref['refined_fraction']:.4f} of {n * hw}) through the reference-order chain "
              f"(codebook_encode_wave_kernel, list form: persistent waves; the quantization head and distances of the levels stage 1 proved are skipped): {listed_flop / 1e9:.1f} GFLOP executed")
        stage("codebook_encode_two_stage", lambda: eng.encode_codes(n), "mfma-f32", enc_gflop * 1e9 * n, "TFLOP/s", F32_MFMA_PEAK_TFLOPS, 3,
              f"both stages + the counter reset, as the step runs them: {n} x {enc_gflop} GFLOP in the REFERENCE's op order are replaced by "
              f"{cand_ops / 1e9:.1f} GOP int8 + {listed_flop / 1e9:.1f} GFLOP fp32 -- `achieved` here is reference-equivalent TFLOP/s, NOT a roofline "
              f"fraction (the stages above carry those)")
        stages["codebook_encode_two_stage"].update({"frac": None, "refined_cells": ref["refined"], "refined_fraction": round(ref["refined_fraction"], 5),
                                                    "first_flagged_at_level": ref["first_flagged_at_level"],
                                                    "reference_order_gflop": round(enc_gflop * n, 2), "executed_gop_int8": round(cand_ops / 1e9, 2),
                                                    "executed_gflop_fp32": round(listed_flop / 1e9, 2)})
        eng.encode_mode = "exact"
    stage("codebook_encode_f32", lambda: eng.encode_codes(n), "mfma-f32", enc_gflop * 1e9 * n, "TFLOP/s",
          F32_MFMA_PEAK_TFLOPS, 1, f"{n} x {enc_gflop} GFLOP (11 chained 256-wide GEMMs per cell, reference op order)"
          + (" -- EVERY cell through the chain (encode_mode 'exact'): not what the step runs, timed beside it" if two_stage else ""))
    if two_stage:
        eng.encode_mode = "two_stage"
        stages["codebook_encode_f32_every_cell"] = stages.pop("codebook_encode_f32")
        eng.encode_codes(n)                              # (leave the workspace as the step does)
    # every launch of the shrinker and the deblocks by itself: candidates for the dominant kernel
    per_layer = {}
    for p_ in sh:
        nm = p_[1].name
        us = event_time_us(lambda: eng.run_plan(n, only=lambda k, l, nm=nm: k == "conv" and l.name == nm), iters)
        per_layer[nm] = (us, 2.0 * p_[7])
    codes = eng._workspace(n)["codes"]
    pw = full["pairwise_t_matrix"].contiguous()
    import ctypes as C
    general = {}                                      # the multi-agent path (decode + warp + attention, then the GEMM heads): not what single-agent scenes run

    def stage_into(dst, name, fn, bound, work, unit, peak, launches, note):
        keep = stages.get(name)
        stage(name, fn, bound, work, unit, peak, launches, note)
        dst[name] = stages.pop(name)
        if keep is not None:
            stages[name] = keep
    heads_macs = hw * 256 * (eng.heads.cout + (eng.heads_single.cout if eng.heads_single is not None else 0))
    by_tables = getattr(eng, "table_heads", None) is not None and getattr(eng, "single_agent_tables", False)
    if by_tables:
        ct = eng.heads.cout + (eng.heads_single.cout if eng.heads_single is not None else 0)
        stage("all_heads_by_tables", lambda: eng._table_heads_out(codes, n), "hbm", n * (3 * hw + ct * hw * 4), "GB/s", HBM_PEAK_GBS, 1,
              f"what a single-agent scene runs for a7-a11 (round 4): AttFusion over ONE agent is the identity, so cls | reg | dir and the *_single heads are "
              f"three table rows per cell (qv2x_table_heads_f32, {ct} channels, tables in LDS): {n} x ({3 * hw / 1e3:.1f} KB codes read + "
              f"{ct * hw * 4 / 1e6:.1f} MB of predictions written); replaces decode_warp_attfuse + heads_f32 below for these scenes")
    fused = torch.empty((n, hw, 256), dtype=torch.float32, device=eng.dev)

    def fuse_all():                                  # as DeployedModel.forward does for multi-agent scenes: every scene of the batch in one launch
        eng.fuse_scenes(C.c_void_p(codes.data_ptr()), hw, n * hw, None, pw, [f * hw for f in range(n)], [1] * n, fused)
    dst = general if by_tables else stages
    stage_into(dst, "decode_warp_attfuse", fuse_all, "hbm", n * (3 * hw + 3 * 128 * 1024 + hw * 1024), "GB/s", HBM_PEAK_GBS, 1,
               f"{n} x ({3 * hw / 1e3:.1f} KB codes + 384 KiB LUT read, {hw * 1024 / 1e6:.1f} MB fp32 fused map written), one launch for the batch's scenes")
    stage_into(dst, "heads_f32", lambda: eng._heads_pair(fused, n, codes, n), "mfma-f32", n * 2.0 * heads_macs, "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
               2 if getattr(eng, "single_by_tables", False) else 1,
               f"{n} x {heads_macs / 1e9:.3f} GMAC: the {eng.heads.cout}-channel heads on the fused map + the {eng.heads_single.cout if eng.heads_single is not None else 0}-channel "
               f"*_single heads on the decoded own feature (three table rows per cell, no GEMM); the fused-map kernel multiplies 96 padded columns")
    if by_tables:
        general["note"] = ("the GENERAL path of a7-a11 (scenes of 2+ agents; every rank of an N-GPU run), timed here on the same single-agent batch for "
                           "reference -- not part of this line's step")
    # ---- `roofline`: the DOMINANT kernel by time of one launch, live --------------------------------------------------------------------
    cands = []                                       # (us per launch, kernel, bound, work per launch, unit, peak, algorithmic bytes, pmc key)
    if two_stage:
        s2 = stages["codebook_encode_listed_f32"]
        rc = stages["codebook_encode_two_stage"]["refined_cells"]
        cands.append((s2["us_per_batch"], "codebook_encode_wave_kernel<.., LIST> (f32 MFMA v_mfma_f32_32x32x2_f32; a wave per 32 listed cells, persistent)",
                      "mfma", stages["codebook_encode_two_stage"]["executed_gflop_fp32"] * 1e9, "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
                      rc * 256 + rc * eng.levels + rc * 4 + sum(int(bl.numel()) * 4 for bl in eng.level_blobs), "encode_listed"))
        s1 = stages["codebook_encode_candidates_i8"]
        cands.append((s1["us_per_batch"], "encode_candidates_kernel (int8 MFMA v_mfma_i32_32x32x32_i8; a wave per 128 cells)", "mfma",
                      2.0 * 3 * 256 * eng.enc_levels * eng.kc * n * hw, "TOP/s", INT8_MFMA_PEAK_TOPS,
                      n * hw * 256 + n * hw * eng.levels + eng.enc_levels * eng.kc * 256 * 3, "encode_candidates"))
    else:
        e_ = stages["codebook_encode_f32"]
        cands.append((e_["us_per_batch"], "codebook_encode_wave_kernel (f32 MFMA v_mfma_f32_32x32x2_f32; a wave per 32 cells)", "mfma",
                      enc_gflop * 1e9 * n, "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
                      n * hw * 256 + n * hw * eng.levels + sum(int(bl.numel()) * 4 for bl in eng.level_blobs), "encode"))
    for p_ in sh:
        us, ops = per_layer[p_[1].name]
        cin = p_[1].w.shape[1] // 9
        cands.append((us, f"conv3x3_i8_wide_kernel ({p_[1].name}: 3x3 {cin}->{p_[1].cout}, int8 MFMA v_mfma_i32_32x32x32_i8, halo patch)", "mfma", ops, "TOP/s",
                      INT8_MFMA_PEAK_TOPS, n * hw * (cin + p_[1].cout) + int(p_[1].w.numel()), "conv_" + p_[1].name))
    de_s = stages["backbone_deconvs_f32"]
    cands.append((de_s["us_per_batch"], "deconv_ps_batch_kernel (f32 MFMA: the three deblocks of the batch in one launch)", "mfma",
                  sum(2.0 * p_[7] for p_ in de), "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
                  # every deblock's input map read once, its slice of the concat written once, the fp32 weights once
                  sum(n * p_[3] * p_[4] * p_[1].cin + n * p_[3] * p_[4] * p_[1].s * p_[1].s * p_[1].cout + int(p_[1].w.numel()) * 4 for p_ in de), "deconv"))
    cands.sort(key=lambda c: -c[0])
    us, kname, bound, work, unit, peak, alg, key = cands[0]
    ach = work / (us * 1e-6) / 1e12
    traffic, note = None, "no PMC profile committed for this kernel in this round yet"
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dominant.json")))     # the latest round's PMC passes, per kernel key
    if found:
        with open(found[-1]) as f:
            j = json.load(f).get(key)
        if j and float(j.get("units_per_launch") or 0) > 0:
            # stored per unit of the kernel's work (agent-frames for the dense kernels, listed cells for stage 2) and scaled to this run's launch
            units = float(stages["codebook_encode_two_stage"]["refined_cells"]) if key == "encode_listed" else float(n)
            traffic = int(round(float(j["traffic_bytes_per_launch"]) / float(j["units_per_launch"]) * units))
            note = f"profiles/{os.path.basename(found[-1])}[{key}]: {j['traffic_bytes_per_launch']} B per launch of {j['units_per_launch']} {j.get('unit', 'units')}, scaled to this launch; " + j.get("note", "")
    roof = {"bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_note": "STORED figure, not measured by this run: " + note,
            "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(traffic / alg, 3) if (traffic and alg) else None,
            "kernel": kname + " -- the dominant kernel by time of one launch", "launches_per_batch": 1,
            "avg_launch_us": round(us, 2), "agent_frames_per_launch": n,
            "algorithmic_gop_per_launch": round(work / 1e9, 2),
            "share_of_batch_time": None,
            "runners_up": [{"kernel": c[1].split(" (")[0], "us": round(c[0], 1), "frac": round(c[3] / (c[0] * 1e-6) / 1e12 / c[5], 4)} for c in cands[1:4]]}
    step_stages = [k for k in stages if k not in ("codebook_encode_f32_every_cell", "codebook_encode_candidates_i8", "codebook_encode_listed_f32")]
    total = sum(stages[k]["us_per_batch"] for k in step_stages if "us_per_batch" in stages[k])
    roof["share_of_batch_time"] = round(us / total, 3)
    if general:
        stages["general_path_multi_agent"] = general
    int8_us = stages["backbone_convs_i8"]["us_per_batch"] + stages["shrinker_convs_i8"]["us_per_batch"]
    int8_ops = sum(2.0 * p[7] for p in bb + sh)
    stages["int8_conv_stack_total"] = {"bound": "mfma-i8", "us_per_batch": round(int8_us, 2), "achieved": round(int8_ops / (int8_us * 1e-6) / 1e12, 1),
                                       "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": round(int8_ops / (int8_us * 1e-6) / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                                       "work": "backbone convs + shrinker (the north_star's int8-MFMA fraction)"}
    return roof, stages


def same_scene_scenes(world, batch=32):
    """scenes per HIP graph of the one-GPU same-scene denominator: the bench's batch for the V2X-Real workloads (N <= 4), 32 agent-frames'
    worth of scenes on the OPV2V grid (N = 8: 4 scenes x 8 agents -- 256 agent-frames of the 512 x 512 grid would be 30 GB of maps)"""
    return batch if world <= 4 else max(1, 32 // world)


def same_scene_one_gpu(eng, world, device, scenes, iters=10):
    """The workload of the N-GPU line (``workload_for(world)``: the same grid, layout and sweeps -- ``frame_batch`` seeds them by agent) WHOLLY on
    one GPU: ``scenes`` scenes of ``world`` agents per HIP graph -- a1-a6 for every agent, a7-a11 once per scene with agent 0 as the ego
    (the reference's single-ego output).  Its fused frames/s is the DENOMINATOR of the scaling figure north_star asks for
    ("matched detection output"): ``bench.py --gpus N --ego-only`` produces the same ego-0 frames on N GPUs.  The engine must be the
    workload's (V2X-Real mc model for N <= 4, the OPV2V single-class model above)."""
    global SHAPE, N_POINTS
    import torch
    wl = workload_for(world)
    keep = (SHAPE, N_POINTS)
    SHAPE, N_POINTS = wl["shape"], wl["n_points"]
    try:
        _, full, _, _ = frame_batch(world, 0, scenes, device, layout=wl["layout"], max_cav=wl["max_cav"])
    finally:
        SHAPE, N_POINTS = keep
    rep = eng.capture(full)
    for _ in range(3):
        rep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        rep()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    del rep
    return {"agents_per_scene": world, "scenes_per_graph": scenes, "agent_frames_per_graph": scenes * world, "ms_per_graph": round(ms, 3),
            "fused_frames_per_s": round(scenes / ms * 1e3, 1), "agent_frames_per_s": round(scenes * world / ms * 1e3, 1),
            "workload": wl["workload"], "baseline_config_index": wl["index"]}


def multi_agent_line(eng, device):
    """BASELINE configs[2] / [3] on ONE GPU (every agent of a scene on this device): scenes of 2 and of 4 agents, 8 agent-frames per
    HIP graph, one graph at a time.  A fused frame = one ego-view detection output of one scene."""
    import torch
    out = {}
    for agents in (2, 4):
        scenes = 8 // agents
        rep = eng.capture(frame_batch(agents, 0, scenes, device)[1])
        for _ in range(3):
            rep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            rep()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        out[f"{agents}_agents"] = {"scenes_per_graph": scenes, "ms_per_graph": round(ms, 3), "fused_frames_per_s": round(scenes / ms * 1e3, 1),
                                   "agent_frames_per_s": round(8 / ms * 1e3, 1)}
        del rep
    out["note"] = ("V2X-Real grid, synthetic sweeps, line (2) / ring (4) layout; a1-a6 per agent, decode + warp + attention over the scene's agents, "
                   "heads on the fused map + *_single heads; not the headline metric (that is the single-agent configuration)")
    # the N-GPU lines' workloads wholly on ONE GPU at the bench's batch: the denominators of `scaling_vs_one_gpu_same_scene`
    same = {}
    for agents in (2, 4):
        try:
            same[f"{agents}_agents"] = same_scene_one_gpu(eng, agents, device, same_scene_scenes(agents))
        except Exception as e:                                             # an extra must not cost the line
            same[f"{agents}_agents"] = {"error": repr(e)[:300]}
    same["note"] = ("BASELINE configs[2] / [3] (and configs[4] under `8_agents_opv2v`, filled by the world-8 rehearsal's engine) with EVERY agent of a scene "
                    "on this GPU, 32 scenes per HIP graph (OPV2V: 4 scenes = 32 agent-frames), ego = agent 0: fused frames/s.  "
                    "`bench.py --gpus N` divides its ego-only figure (the same ego-0 frames, agents sharded one per GPU) by a run of this on rank 0: "
                    "`scaling_vs_one_gpu_same_scene`")
    out["same_scene_b32"] = same
    return out


def codebook_seg_line(full, B, device):
    """The codebook setting of six of the reference's ten codebook yamls -- ``seg_num: 2, dict_size: 256`` (every OPV2V / DAIR-V2X codebook
    config: hypes_yaml/opv2v/Codebook/Pyramid/pyramid_stage2_model.yaml:96-97, v2x_real/Codebook/Attfuse/lidar_attfuse_stage2.yaml:114-115)
    -- on this run's V2X-Real workload: the same step (B frames per graph, one graph at a time) and the encode kernel alone.  Two segments
    of 128 dims, 256 codes each: six code planes per agent-frame on the wire (211 200 B) and a distance GEMM of 512 columns per level
    (the extended codebook, include/qv2x.h) where (1, 128) has 128 -- 32.3 GMAC per agent-frame instead of 21.9 as executed."""
    import copy
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    model = train_utils.create_model(copy.deepcopy(synth.make_hypes(SHAPE, multiclass=True, dict_size=256, seg_num=2))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    calib = synth.scene_to_torch(synth.make_scene(SHAPE, n_agents=1, seed=3, n_points=N_POINTS))
    eng = deploy(state=export_ptq_state(inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib])))
    rep = eng.capture(full)
    for _ in range(3):
        rep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        rep()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    us = event_time_us(lambda: eng.encode_codes(B), 5)
    hw = eng.fh * eng.fw
    # per agent-frame as executed: three 256 x 256 heads per level (two on the last) + the distances to a segment's codes over the segment's
    # dims only (round 5: the wave form walks the extended codebook's diagonal blocks -- seg_num x [dict_size x 256 / seg_num])
    macs = hw * 3 * (3 * 65536 + eng.segs * eng.kc * (256 // eng.segs)) - hw * 65536
    return {"seg_num": eng.segs, "dict_size": eng.kc, "code_planes": eng.levels, "wire_bytes_per_agent_frame": eng.levels * hw,
            "frames_per_s": round(B / ms * 1e3, 1), "ms_per_step": round(ms, 4), "frames_per_step": B, "batches_in_flight": 1,
            "encode_us_per_batch": round(us, 1), "encode_tflops_as_executed": round(2.0 * macs * B / us / 1e6, 1),
            "heads_by_tables": eng.table_heads is not None,
            "note": "the same V2X-Real single-agent batch with the (2, 256) codebook; ONE graph in flight (the headline runs two); six code planes: "
                    "the single-agent table look-up reads its 565 KB of tables from global memory (round 5; the LDS form holds four planes)"}


def pyramid_model_line(device):
    """SURVEY.md §8(f) rank 3, reported beside the headline: the HEAL Pyramid-fusion model (2 agents per scene, V2X-Real grid) on its
    own engine -- one frame as a HIP graph, and batches of 4 scenes."""
    import copy
    import numpy as np
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    model = train_utils.create_model(copy.deepcopy(synth.make_pyramid_hypes(SHAPE))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    calib = synth.scene_to_torch(synth.make_scene(SHAPE, n_agents=1, seed=3, n_points=N_POINTS))
    eng = deploy(state=export_ptq_state(inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib])), device=device)

    def batch(frames, agents=2):
        scenes = [synth.make_scene(SHAPE, n_agents=agents, seed=3 + f, n_points=N_POINTS) for f in range(frames)]
        parts = []
        for f, sc in enumerate(scenes):
            part = {k: v.copy() for k, v in sc["inputs_m1"].items()}
            part["voxel_coords"][:, 0] += f * agents
            parts.append(part)
        return {"inputs_m1": {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).to(device) for k in parts[0]},
                "agent_modality_list": ["m1"] * (agents * frames), "record_len": torch.full((frames,), agents, dtype=torch.int64),
                "pairwise_t_matrix": torch.from_numpy(np.concatenate([sc["pairwise_t_matrix"] for sc in scenes])).to(device)}
    out = {}
    for frames in (1, 4):
        rep = eng.capture(batch(frames))
        us = event_time_us(rep, 30)
        out["ms_per_frame" if frames == 1 else f"ms_per_frame_batch{frames}"] = round(us / frames / 1e3, 4)
    out["frames_per_s_batch4"] = round(1e3 / out["ms_per_frame_batch4"], 1)
    # ---- roofline entries of the Pyramid model: one 2-agent frame, the two halves as HIP graphs and the launches counted ------
    try:
        dd = batch(1)
        inv = eng.work_inventory(2, 1)
        hw = eng.fh * eng.fw
        codes = eng.encode_features(dd["inputs_m1"], 2).clone()
        pw = dd["pairwise_t_matrix"].to(torch.float64).contiguous()
        enc_us = event_time_us(_graph_of(lambda: eng.encode_features(dd["inputs_m1"], 2)), 30)
        dec_us = event_time_us(_graph_of(lambda: eng.decode_features(codes, hw, 2 * hw, [2], pw)), 30)
        counted = {"n": 0}

        class _Count:                                                       # every qv2x_* call of one eager forward = one kernel launch (or a short fixed group)
            def __init__(self, lib):
                self._lib = lib

            def __getattr__(self, name):
                f = getattr(self._lib, name)
                if not name.startswith("qv2x_") or name.endswith("_floats") or name.endswith("_bytes"):
                    return f

                def g(*a, **k):
                    counted["n"] += 1
                    return f(*a, **k)
                return g
        real = eng.lib
        eng.lib = _Count(real)
        try:
            eng(dd)
            torch.cuda.synchronize()
        finally:
            eng.lib = real
        i8_ego = sum(l["int8_macs"] for l in inv["levels"]) + inv["shrink_int8_macs"]
        f32_ego = sum(l["deblock_f32_macs"] for l in inv["levels"]) + inv["heads_f32_macs"]
        agent_ops = 2.0 * inv["agent_int8_macs"]
        out["roofline_stages"] = {
            "agent_side_encode_features": {"us": round(enc_us, 1), "int8_gop": round(agent_ops / 1e9, 2), "f32_encode_gflop": round(2.0 * inv["encode_f32_macs"] / 1e9, 2),
                                           "achieved_int8_TOPs": round(agent_ops / (enc_us * 1e-6) / 1e12, 1), "peak": INT8_MFMA_PEAK_TOPS,
                                           "frac_int8": round(agent_ops / (enc_us * 1e-6) / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                                           "work": "PFN + scatter, the agent's 3 ResNet blocks (int8 MFMA), the 64-wide codebook encode (f32 MFMA) for 2 agents"},
            "ego_side_decode_features": {"us": round(dec_us, 1), "int8_gop": round(2.0 * i8_ego / 1e9, 2), "f32_gflop": round(2.0 * f32_ego / 1e9, 2),
                                         "hbm_mb": round((inv["decode_bytes"] + sum(l["fuse_bytes"] for l in inv["levels"])) / 1e6, 2),
                                         "achieved_int8_TOPs": round(2.0 * i8_ego / (dec_us * 1e-6) / 1e12, 1), "peak": INT8_MFMA_PEAK_TOPS,
                                         "frac_int8": round(2.0 * i8_ego / (dec_us * 1e-6) / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                                         "work": "decode, 16 ResNeXt bottlenecks per agent over three levels, occupancy heads + weighted fusion, deblocks (f32), shrink_conv, heads"},
            "levels": [{"level": i, "blocks": l["blocks"], "map": [l["h"], l["w"]], "planes": l["planes"], "int8_gop": round(2.0 * l["int8_macs"] / 1e9, 3),
                        "us_at_int8_peak": round(2.0 * l["int8_macs"] / (INT8_MFMA_PEAK_TOPS * 1e12) * 1e6, 2)} for i, l in enumerate(inv["levels"])],
            "launches_per_frame": counted["n"],
            "note": "a 2-agent frame is ~%d launches on maps of 100 x 352 .. 25 x 88 cells: %.2f GOP of int8 work would take %.1f us at the MFMA peak -- the frame is bound by "
                    "launch latency and small grids, not by a pipe (DESIGN.md 3)" % (counted["n"], (agent_ops + 2.0 * i8_ego) / 1e9,
                                                                                      (agent_ops + 2.0 * i8_ego) / (INT8_MFMA_PEAK_TOPS * 1e12) * 1e6)}
    except Exception as e:                                                 # an extra must not cost the line
        out["roofline_stages"] = {"error": repr(e)[:300]}
    out["note"] = ("heter_pyramid_collab_codebook_mc_encdec under W8A8, 2 agents per scene: agent side (3 ResNet blocks, 64-wide codebook "
                   "encode) + ego side (decode, 16 ResNeXt bottlenecks per agent, occupancy-weighted fusion, deblocks, shrink, heads)")
    return out


def points_to_boxes_line(state, device):
    """The deployment chain around the model, as ONE HIP graph per batch: LiDAR sweeps -> pillars (qv2x_voxelize_f32, replacing the
    reference's CPU spconv voxelizer, pre_processor/sp_voxel_preprocessor.py:54-85) -> the model (a1-a11) -> boxes (qv2x_postprocess_f32:
    VoxelPostprocessor3Heads.post_process, data_utils/post_processor/voxel_postprocessor_3heads.py:318-478; what
    tools/inference_utils.py:201-225 calls per frame).  No host read-back inside the graph: every sweep hands a fixed number of pillar rows
    on (unused rows carry batch index -1) and the box count stays on the device.  Not part of ``value``."""
    import numpy as np
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    from quantv2x_amd.plugin.data_utils.post_processor.voxel_postprocessor import gpu_post_process
    from quantv2x_amd.voxelizer import GpuVoxelizer
    lidar, vox, max_vox, _ = synth.SHAPES[SHAPE]
    gw, gh, _ = synth.grid_size(lidar, vox)
    eng = deploy(state=state)
    vz = GpuVoxelizer(lidar, vox, 32, max_vox)
    pp = build_postprocessor(synth.mc_postprocess_params(lidar, gw, gh), train=False)
    all_anchors, _ = pp.generate_anchor_box()
    a = torch.as_tensor(np.array(all_anchors)).to(torch.float32).permute(1, 2, 0, 3, 4).contiguous()     # (H, W, class, anchor, 7)
    anchors_dev, per_cell = a.reshape(-1, 7).to(device), int(a.shape[2] * a.shape[3])
    cap = 40960                                       # pillar rows handed on per sweep (the synthetic sweeps fill ~27k; the yaml's limit is 70k)
    out = {"pillar_rows_per_sweep": cap}
    for frames in (1, 8):
        sweeps = [torch.from_numpy(synth.make_points(lidar, N_POINTS, 3000 + f)).to(device) for f in range(frames)]
        pairwise = torch.eye(4, dtype=torch.float64).reshape(1, 1, 1, 4, 4).repeat(frames, 5, 5, 1, 1).to(device)
        rl = torch.ones(frames, dtype=torch.int64)

        def run():
            inp = vz.fixed(sweeps, cap)
            o = eng({"inputs_m1": inp, "agent_modality_list": ["m1"] * frames, "record_len": rl, "pairwise_t_matrix": pairwise})
            res = []
            for f in range(frames):
                res.append(gpu_post_process(pp, o["cls_preds"][f:f + 1], o["reg_preds"][f:f + 1], None, anchors_dev, torch.eye(4),
                                            anchors_per_cell=per_cell, num_classes=int(o["cls_preds"].shape[1] // per_cell), num_bins=0,
                                            dir_offset=0.0, rng=pp.gt_range, range_xy_only=True, max_extent=100.0, z_lim=(-100.0, 100.0),
                                            max_boxes=1000, sync=False))
            return inp, res
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            inp, res = run()
        graph.replay()
        torch.cuda.synchronize()
        us = event_time_us(graph.replay, 30)
        key = "one_frame" if frames == 1 else f"batch{frames}"
        out[key] = {"ms_per_frame": round(us / frames / 1e3, 4), "frames_per_s": round(frames * 1e6 / us, 1),
                    "pillars": [int(v) for v in inp["voxel_counts"].tolist()][:2], "boxes": [int(r[3].item()) for r in res][:2]}
        if frames == 1:
            lat = []
            for _ in range(30):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t0) * 1e3)
            out[key]["latency_ms_p50"] = round(sorted(lat)[len(lat) // 2], 4)
        del graph
    # ---- the two stages around the model as roofline entries: each alone inside a HIP graph, batch of 8 ----------------
    frames = 8
    sweeps = [torch.from_numpy(synth.make_points(lidar, N_POINTS, 3000 + f)).to(device) for f in range(frames)]
    us_v = event_time_us(_graph_of(lambda: vz.fixed(sweeps, cap)), 30)
    vbytes = frames * (N_POINTS * 16 + cap * (32 * 16 + 16 + 4))       # points read once; every handed-on pillar row written (zero-padded slots too)
    out["stages"] = {"voxelize": {"bound": "hbm", "us_per_batch": round(us_v, 1), "frames": frames, "achieved": round(vbytes / us_v / 1e3, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(vbytes / us_v / 1e3 / HBM_PEAK_GBS, 4),
                                  "work": f"{frames} sweeps x ({N_POINTS} points x 16 B read + {cap} pillar rows x 532 B written: the padded [rows][32][4] fp32 "
                                          f"block the model's contract asks for); a 64-bit key sort over the points, segment heads, scatter"}}
    pairwise = torch.eye(4, dtype=torch.float64).reshape(1, 1, 1, 4, 4).repeat(frames, 5, 5, 1, 1).to(device)
    o = eng({"inputs_m1": vz.fixed(sweeps, cap), "agent_modality_list": ["m1"] * frames, "record_len": torch.ones(frames, dtype=torch.int64),
             "pairwise_t_matrix": pairwise})
    cls, reg = o["cls_preds"].clone(), o["reg_preds"].clone()

    def post():
        return [gpu_post_process(pp, cls[f:f + 1], reg[f:f + 1], None, anchors_dev, torch.eye(4), anchors_per_cell=per_cell,
                                 num_classes=int(cls.shape[1] // per_cell), num_bins=0, dir_offset=0.0, rng=pp.gt_range, range_xy_only=True,
                                 max_extent=100.0, z_lim=(-100.0, 100.0), max_boxes=1000, sync=False) for f in range(frames)]
    res = post()
    torch.cuda.synchronize()
    us_p = event_time_us(_graph_of(post), 30)
    hw = eng.fh * eng.fw
    pbytes = frames * (int(cls.shape[1]) + int(reg.shape[1])) * hw * 4
    out["stages"]["postprocess"] = {"bound": "latency (rotated-IoU rows + the greedy keep sweep, one wave)", "us_per_batch": round(us_p, 1), "frames": frames,
                                    "achieved": round(pbytes / us_p / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(pbytes / us_p / 1e3 / HBM_PEAK_GBS, 4),
                                    "nms_pairs_per_s": round(frames * 1000 * 999 / 2 / us_p * 1e6, 0),
                                    "work": f"{frames} frames x ({int(cls.shape[1])} + {int(reg.shape[1])}) head maps x {hw} cells x 4 B read; sigmoid + box decode + "
                                            f"radix select of the top 1000 scores + one workgroup's sort + rotated IoU of up to 1000 x 999 / 2 candidate pairs + the keep sweep, 64 candidates a step; boxes kept: "
                                            f"{[int(r[3].item()) for r in res][:2]}"}
    out["note"] = ("60k-point synthetic sweeps -> voxelize -> W8A8 model -> sigmoid / decode / rotated NMS (random-weight heads: ~450 boxes "
                   "out of 1000 candidates, the NMS works hard) as one hipGraph; the model body alone is `value` / `latency_ms_p50`")
    return out


def _graph_of(fn):
    """``fn`` captured into a HIP graph after two warm-up calls; returns its replay"""
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    g.keep = keep
    return g.replay


def encode_modes_line(state, full, B, F, steps, device):
    """NOT the headline: the same step (B frames per graph, F graphs in flight) with the codebook encode in its other modes -- ``exact``:
    EVERY cell through the reference-order chain (the configuration of rounds 1-5; identical indices, checked here on the whole batch);
    ``collapsed``: the opt-in approximate form (one fp32 GEMM + an argmin chain; differs at near-ties, tools/bench_collapsed_encode.py)."""
    import torch
    from quantv2x_amd.engine import deploy
    engines = [deploy(state=state) for _ in range(F)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    out = {}
    ref = engines[0]
    default_mode = ref.encode_mode
    want = ref.encode_agents(full["inputs_m1"], B).clone()               # the default mode's indices
    for mode in ("exact", "collapsed"):
        reps = []
        for e, st in zip(engines, streams):
            e.encode_mode = mode
            with torch.cuda.stream(st):
                reps.append(e.capture(full))
        torch.cuda.synchronize()
        for i in range(2 * F):
            with torch.cuda.stream(streams[i % F]):
                reps[i % F]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % F]):
                reps[i % F]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        got = ref.encode_agents(full["inputs_m1"], B)
        torch.cuda.synchronize()
        out[mode] = {"frames_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
                     "index_mismatches_vs_default_mode": int((got != want).sum().item()), "indices_compared": int(want.numel())}
        del reps
    for e in engines:
        e.encode_mode = default_mode
    out["default_mode"] = default_mode
    out["note"] = ("`value` runs the default mode (two_stage where its contract holds: exact by construction).  'exact' = every cell through the eleven "
                   "chained GEMMs (what rounds 1-5 timed); 'collapsed' = opt-in, approximate, not a parity configuration")
    return out


def second_encoder_line(device):
    """SURVEY.md §8 row a13, reported beside the headline: the quantized SECOND encoder (MeanVFE + 12 sparse 3-D convolutions + height
    compression) on one full-size synthetic sweep (0.1 m voxels over the V2X-Real range), as one HIP graph."""
    import torch
    import torch.nn as nn
    from quantv2x_amd import synth
    from quantv2x_amd.engine_second import DeployedSecondEncoder
    from quantv2x_amd.plugin.models.heter_encoders import SECOND
    from quantv2x_amd.plugin.quant import QuantModel, set_act_quantize_params, set_weight_quantize_params
    from quantv2x_amd.ptq_state import export_second_state

    class EncoderOnly(nn.Module):
        def __init__(self, enc):
            super().__init__()
            self.encoder_m1 = enc

        def forward(self, dd):
            return self.encoder_m1(dd, "m1")

    shape = "second_full"
    enc = SECOND(synth.make_second_args(shape)).eval()
    synth.load_state_dict_numpy(enc, synth.make_state_dict(enc.state_dict(), seed=1))
    qm = QuantModel(EncoderOnly(enc), dict(n_bits=8, channel_wise=True, scale_method="minmax"),
                    dict(n_bits=8, channel_wise=False, scale_method="minmax", leaf_param=True)).eval()
    calib = synth.make_second_scene(shape, 1, seed=3, n_points=15000)          # ranges from a thinner sweep: they set values, not time
    set_weight_quantize_params(qm)
    set_act_quantize_params(qm, [{"inputs_m1": {k: torch.from_numpy(v) for k, v in calib.items()}}])
    eng = DeployedSecondEncoder(export_second_state(qm.model.encoder_m1), device, agents=1, max_voxels=synth.SECOND_SHAPES[shape][2])
    sweep = synth.make_second_scene(shape, 1, seed=3, n_points=N_POINTS)
    inp = {k: torch.from_numpy(v).to(device) for k, v in sweep.items()}
    taps = {}
    eng(inp, taps)
    torch.cuda.synchronize()
    sites = [int(taps[f"second/{i}"][2].item()) for i in range(len(eng.layers))]
    gmac = sum(n * ly.K * ly.ci * ly.co for n, ly in zip(sites, eng.layers)) / 1e9
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng(inp)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eng(inp)
    torch.cuda.synchronize()
    us = event_time_us(graph.replay, 30)
    return {"ms_per_sweep": round(us / 1e3, 4), "sweeps_per_s": round(1e6 / us, 1), "voxels": int(sweep["voxel_coords"].shape[0]),
            "active_sites_per_level": [sites[1], sites[4], sites[7], sites[10], sites[11]], "dense_window_gmac": round(gmac, 2),
            "int8_top_s_equivalent": round(2 * gmac / us * 1e3, 1),
            "note": "QuantSECOND under W8A8 (parity against spconv itself unpinned, DESIGN.md 1): rulebooks + gather-GEMM on int8 MFMA; "
                    "dense_window_gmac counts every window offset of every active output, occupied or not"}


def cpu_baseline(state, scenes_np, gpu_check=None, budget_s=20.0, max_frames=12):
    """The CPU oracle (checker) on the same workload, on the host cores of this box: the frames of the bench's own batch, one at a time.
    ``gpu_check(f, oracle_taps) -> (index_mismatches, u8_mismatches)`` compares what the oracle just computed for frame f with what the
    timed HIP graphs left in the workspace (parity gates with every timing) -- the oracle is the checker here, never
    the thing shipped; the comparison is outside the timed interval."""
    from oracle.spec import Oracle
    cores = os.cpu_count() or 1
    orc = Oracle(state)
    orc.forward(scenes_np[0])               # warm-up (also builds the shared object if needed)
    spent, frames, parity = 0.0, 0, {"index_mismatches": 0, "u8_mismatches": 0, "frames_checked": 0}
    while frames < min(max_frames, len(scenes_np)) and spent < budget_s:
        taps = {}
        t0 = time.time()
        orc.forward(scenes_np[frames], taps)
        spent += time.time() - t0
        if gpu_check is not None:
            im, um = gpu_check(frames, taps)
            parity["index_mismatches"] += im
            parity["u8_mismatches"] += um
            parity["frames_checked"] += 1
        frames += 1
    out = {"value": round(frames / spent, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{frames} full single-agent V2X-Real frames of the bench's own batch (whole hot path) through oracle/ in {spent:.1f} s, "
                     f"OpenMP on {cores} threads"}
    return out, (parity if gpu_check is not None else None)


def cpu_baseline_torch(fp_model, qt, sc_np, cores, warmup=3, timed=10):
    """SURVEY.md §8(d) / BASELINE.md §4 protocol: the torch restatement of the reference (the plugin mirror: same modules, same torch
    ops) on the host cores -- fp32, and W8A8 fake-quant (quantize -> dequantize in fp32 around fp32 F.conv2d, the reference's only
    execution mode; quant_layer.py:391-410).  The thread count is swept over {32, 64, all host cores} with one fp32 frame each
    (a count is skipped once the sweep has started to get slower: torch's CPU convolutions lose to oversubscription, 45 s per fp32
    frame on 256 threads measured in round 3), then ``warmup`` + ``timed`` frames per mode at the best count, with the per-stage split
    of BASELINE.md §2 (encoder / backbone / shrinker / codebook / fusion / heads) from forward hooks on the mirror's modules
    (timing protocol of tools/profiler/params_calc.py:48-79)."""
    import torch
    from quantv2x_amd import synth
    dd = synth.scene_to_torch(sc_np)
    out = {"host_cores": cores, "unit": "frames/s", "kind": "torch restatement of the reference (plugin mirror)", "warmup_frames": warmup,
           "timed_frames": timed}

    def run(m):
        with torch.no_grad():
            torch.manual_seed(0)
            m(dd)

    sweep, prev = {}, None
    for th in sorted({min(32, cores), min(64, cores), cores}):
        if prev is not None and len(sweep) >= 2 and prev > 1.15 * min(sweep.values()):
            sweep[th] = None                                  # already getting slower: not worth a 45-s frame
            continue
        torch.set_num_threads(th)
        run(fp_model)
        t0 = time.time()
        run(fp_model)
        prev = sweep[th] = time.time() - t0
    best = min((t, th) for th, t in sweep.items() if t is not None)[1]
    out["thread_sweep_fp32_s_per_frame"] = {str(k): (None if v is None else round(v, 3)) for k, v in sweep.items()}
    out["cores"] = best
    torch.set_num_threads(best)

    stage_of = (("encoder_m1", "encoder"), ("backbone_m1", "backbone"), ("shrinker_m1", "shrinker"), ("codebook", "codebook"),
                ("fusion_net", "fusion"), ("cls_head", "heads"), ("reg_head", "heads"), ("dir_head", "heads"),
                ("cls_head_single", "heads"), ("reg_head_single", "heads"), ("dir_head_single", "heads"))
    for name, m in (("fp32", fp_model), ("w8a8_fake_quant", qt)):
        inner = m.model if hasattr(m, "model") else m
        acc, t_in, hooks = {}, {}, []
        for attr, stage in stage_of:
            mod = getattr(inner, attr, None)
            if mod is None:
                continue
            hooks.append(mod.register_forward_pre_hook(lambda _m, _i, a=attr: t_in.__setitem__(a, time.perf_counter())))
            hooks.append(mod.register_forward_hook(lambda _m, _i, _o, a=attr, st=stage: acc.__setitem__(st, acc.get(st, 0.0) + time.perf_counter() - t_in[a])))
        for _ in range(warmup):
            run(m)
        acc.clear()
        t0 = time.time()
        for _ in range(timed):
            run(m)
        dt = time.time() - t0
        for h in hooks:
            h.remove()
        out[name] = round(timed / dt, 4)
        out[name + "_sample"] = f"{timed} single-agent V2X-Real frames in {dt:.1f} s after {warmup} warm-up frames, {best} threads"
        out[name + "_ms_per_stage"] = {k: round(v / timed * 1e3, 1) for k, v in acc.items()}
    return out


def self_launch(n, argv, check_devices=True):
    """``python bench.py --gpus N`` without a launcher: start ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
    FRESH CHILD (this process has not touched a GPU: counting devices does not initialise one, and a process that did must never
    be replaced by exec), pass its output through (rank 0's JSON line is the last line of stdout) and return its exit code.
    torchrun picks the rendezvous port itself (--standalone: a c10d store on a free port of 127.0.0.1 -- no bind / close / re-bind race
    with another job on the node).  Reference analogue of the launch: opencood/tools/train_ddp.py:46-106,
    tools/multi_gpu_utils.py:16-38 (env:// rendezvous)."""
    import subprocess
    if check_devices:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but this node has {have} GPU(s); not printing a line for a job that did not run", file=sys.stderr)
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n),
           os.path.abspath(__file__)] + list(argv)
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run_ranks(rank, world, args):
    """The launcher path without GPUs: gloo group, one all-reduce, rank 0 prints a line carrying the world size and the workload
    (BASELINE config, shape, heads, max_cav) a real run at this world size would execute."""
    import torch
    import torch.distributed as dist
    from quantv2x_amd import synth
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ok = float(t.item()) == world * (world + 1) / 2
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        wl = workload_for(world)
        lidar, vox, _, _ = synth.SHAPES[wl["shape"]]
        gw, gh, _ = synth.grid_size(lidar, vox)
        print(json.dumps({"metric": "launcher dry run (no GPU work)", "n_gpus": world, "dry_run": True, "ranks_joined": ok,
                          "steps": args.steps, "warmup": args.warmup, "ego_only": bool(args.ego_only),
                          "config": {"workload": wl["workload"], "baseline_config_index": wl["index"], "shape": wl["shape"],
                                     "voxel_grid": [gw, gh], "feature_map": [gh // 2, gw // 2], "layout": wl["layout"], "max_cav": wl["max_cav"],
                                     "multiclass_heads": wl["multiclass"], "points_per_agent": wl["n_points"],
                                     "agents_per_frame": world, "batch_per_rank": args.batch}}), flush=True)
    return 0 if ok else 1


def rehearse_line(W, B, steps, warmup, device, engine=None, rank=0):
    """Rank ``rank``'s step of the W-GPU line on one GPU (``AgentShardedModel(emulate_world=W)``): the own agent's B frames encoded, the own payload
    copied into every agent slot with the agents' poses beside it (a device copy where the all-gather would be), then the pairwise matrices, the
    fusion of W agents and the heads -- its two stages timed with HIP events.  ``engine``: a deployed model of the workload's shape to reuse
    (the N = 1 line's own engine serves W = 2..4: same V2X-Real model); None builds (calibrates) the workload's model."""
    global SHAPE, N_POINTS
    import numpy as np
    import torch
    from quantv2x_amd.dist import AgentShardedModel
    wl = workload_for(W)
    keep = (SHAPE, N_POINTS)
    SHAPE, N_POINTS = wl["shape"], wl["n_points"]
    try:
        eng = engine
        if eng is None:
            _, eng, _, _ = build_engine(max(1, min(32, os.cpu_count() or 8)), multiclass=wl["multiclass"])
        _, _, mine, poses = frame_batch(W, rank, B, device, layout=wl["layout"], max_cav=wl["max_cav"], own_only=True)
        n_points = N_POINTS
    finally:
        SHAPE, N_POINTS = keep
    pose_t = torch.from_numpy(np.stack(poses)).to(device)
    tables = eng.single_agent_tables
    sh = AgentShardedModel(eng, frames=B, max_cav=wl["max_cav"], emulate_world=W, emulate_poses=pose_t, emulate_rank=rank)
    out = sh.forward(mine, pose_t[rank])
    torch.cuda.synchronize()
    pre, post = sh.stage_graphs()

    def step():
        sh.forward(mine, pose_t[rank])
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pre_us = event_time_us(pre.replay, max(5, steps // 5))
    post_us = event_time_us(post.replay, max(5, steps // 5))
    eng.single_agent_tables = tables
    hw = eng.fh * eng.fw
    return {"metric": "frames/sec of ONE rank's step (rehearsal of an N-GPU run on one GPU; NOT a multi-GPU measurement)",
            "value": round(B * steps / dt, 2), "unit": "frames/s", "n_gpus": 1, "rehearsal_of_n_gpus": W, "rehearsed_rank": rank, "ego": rank, "steps": steps,
            "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "dtype": "i8", "data": "synthetic",
            "config": {"workload": wl["workload"], "baseline_config_index": wl["index"], "grid": wl["grid"], "agents_per_frame": W,
                       "max_cav": wl["max_cav"], "layout": wl["layout"], "points_per_agent": n_points, "batch_per_rank": B,
                       "heads": "multi-class (mc, 72 channels)" if wl["multiclass"] else "single-class (20 channels)",
                       "pillars_per_step": int(mine["voxel_features"].shape[0]), "wire_bytes_per_agent_frame": eng.levels * hw,
                       "launch": "hipGraph (a1-a6 on the own agent's frames) -> the own payload copied into every agent slot, the agents' poses "
                                 "written beside it (stands in for the all-gather) -> hipGraph (pairwise matrices, a7-a11 over W agents)"},
            "stage_us": {"pre_a1_to_a6": round(pre_us, 1), "post_a7_to_a11": round(post_us, 1)},
            "note": f"what every rank of `bench.py --gpus {W}` executes per step, with the collective replaced by a device copy; with a free "
                    f"link the {W}-GPU line would read {W} x value (every rank the ego of its own view)",
            "output_shapes": {k: list(v.shape) for k, v in out.items() if hasattr(v, "shape")},
            "_engine": eng}                                  # (for a second rank of the same world: popped by the callers, never printed)


def rehearse(args, device):
    """--rehearse-world W (see its help)"""
    import torch
    torch.cuda.set_device(device)
    line = rehearse_line(args.rehearse_world, max(1, args.batch), args.steps, args.warmup, device, rank=args.rehearse_rank)
    line.pop("_engine")
    print(json.dumps(line), flush=True)
    return 0


def sharded_world1_line(state, B, device, steps=20):
    """The N > 1 code path at world 1 inside the default line: ``AgentShardedModel`` over the C ABI's own RCCL communicator
    (``qv2x_allgather_codes``) with the all-gather captured INSIDE the step's HIP graph (pre + collective + post = one replay per step), and the
    two-graph form over torch.distributed's communicator beside it.  The post stage is the GENERAL a7-a11 path (the single-agent table
    shortcut is switched off: a real rank never takes it).  stdout is parked on stderr meanwhile: RCCL prints a banner from C."""
    import torch
    import torch.distributed as dist
    from quantv2x_amd.dist import AgentShardedModel
    from quantv2x_amd.engine import deploy
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    out = {}
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
        wl = workload_for(1)
        _, _, mine, poses = frame_batch(1, 0, B, device, layout=wl["layout"], max_cav=wl["max_cav"])
        pose = torch.from_numpy(poses[0]).to(device)
        for tag, link, graph_link in (("rccl_one_graph", "rccl", True), ("torch_two_graphs", "torch", False)):
            eng = deploy(state=state)
            eng.single_agent_tables = False
            sh = AgentShardedModel(eng, frames=B, link=link, max_cav=wl["max_cav"], graph_link=graph_link)
            sh.forward(mine, pose)
            torch.cuda.synchronize()
            for _ in range(3):
                sh.forward(mine, pose)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sh.forward(mine, pose)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[tag] = {"frames_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps}
            sh.close()
            del sh, eng
        out["note"] = ("world 1, ONE batch in flight, general a7-a11 post stage: `--force-sharded --link rccl --graph-link` (the all-gather captured in "
                       "the step's graph) and `--force-sharded --link torch` (pre graph, eager all-gather, post graph); not a multi-GPU measurement")
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return out


def main():
    global SHAPE, N_POINTS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="frames per step (per rank); 32: every persistent conv workgroup owns several items in a row (DESIGN.md 5)")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight per rank (streams / engines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra fields (fp32 path, multi-agent, Pyramid, SECOND, collapsed encode, "
                    "points -> boxes): the line's contract fields, roofline and roofline_stages only (profiling passes)")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path (HIP graphs around the collective) with one rank")
    ap.add_argument("--link", default="torch", choices=["torch", "rccl"], help="N>1 collective: torch.distributed or qv2x_allgather_codes")
    ap.add_argument("--graph-link", action="store_true", help="with --link rccl: the all-gather is captured INSIDE the step's one HIP graph "
                    "(pre + qv2x_allgather_codes + post = one replay per step) instead of sitting between two graphs")
    ap.add_argument("--ego-only", action="store_true", help="N>1: only rank 0 fuses (SURVEY 8(e)(i), the parity configuration: its output equals "
                    "the single-process model's); `value` then counts rank 0's frames only.  Without the flag every rank is the ego of its own "
                    "view (8(e)(ii)) and the ego-only figure is reported beside it as `ego_only`")
    ap.add_argument("--rehearse-world", type=int, default=0, metavar="W",
                    help="ONE GPU plays rank 0 of a W-GPU run: the workload BASELINE.json names for W (workload_for), the own agent's frames "
                         "encoded, every other agent slot of the gathered payload filled with the own code planes + that agent's pose, the "
                         "fusion of W agents and the heads -- the per-rank step of the W-GPU line with its true shapes, WITHOUT the link. "
                         "Prints its own line (n_gpus 1, rehearsal_of_n_gpus W): not a multi-GPU measurement")
    ap.add_argument("--rehearse-rank", type=int, default=0, metavar="R",
                    help="with --rehearse-world W: the rank played (its own sweeps, its pose, ego = R: what ranks 1 .. W-1 of the default "
                         "every-rank-is-its-own-ego mode execute); default 0")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="launcher check (CPU, gloo): every rank joins the group, rank 0 prints a line with n_gpus = world size; no GPU work")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_run_ranks):
        # a bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) and relay rank 0's line
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], check_devices=not args.dry_run_ranks))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")
    if args.dry_run_ranks:
        raise SystemExit(dry_run_ranks(rank, world, args))
    if torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {args.gpus}: this node has {torch.cuda.device_count()} GPU(s)")
    if args.graph_link and args.link != "rccl":
        raise SystemExit("--graph-link needs --link rccl (the C ABI's own communicator: qv2x_allgather_codes is captured into the step's graph)")
    if args.rehearse_world:
        if world != 1 or args.rehearse_world < 2 or args.rehearse_world > 8:
            raise SystemExit("--rehearse-world W: one process, 2 <= W <= 8")
        if not 0 <= args.rehearse_rank < args.rehearse_world:
            raise SystemExit("--rehearse-rank R: 0 <= R < W")
        raise SystemExit(rehearse(args, torch.device("cuda", local)))
    wl = workload_for(world)
    SHAPE, N_POINTS = wl["shape"], wl["n_points"]
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    sharded_mode = world > 1 or args.force_sharded
    saved_stdout = None
    if sharded_mode:
        # RCCL writes a version banner to the C stdout of every rank: until the line is ready, fd 1 is stderr, so that the JSON line is the
        # ONLY thing this job puts on stdout
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    cores = os.cpu_count() or 8
    B = max(1, args.batch)
    state, eng, fp_model, qt = build_engine(max(1, min(32, cores // max(world, 1))), multiclass=wl["multiclass"])
    sc_np, full, mine, poses = frame_batch(world, rank, B, device, layout=wl["layout"], max_cav=wl["max_cav"], own_only=world > 1)
    ego_line = None

    if not sharded_mode:
        from quantv2x_amd.engine import deploy
        F = max(1, args.inflight)
        engines = [eng] + [deploy(state=state) for _ in range(F - 1)]
        streams = [torch.cuda.Stream() for _ in range(F)]
        reps = []
        for e, st in zip(engines, streams):
            with torch.cuda.stream(st):
                reps.append(e.capture(full))            # one HIP graph = one batch of B frames
        torch.cuda.synchronize()
        it = [0]

        def step():
            i = it[0] % F
            it[0] += 1
            with torch.cuda.stream(streams[i]):
                reps[i]()
        launch = f"hipGraph replay of a {B}-frame batch, {F} batches in flight on {F} streams"
        frames_per_step = B
    else:
        from quantv2x_amd.dist import AgentShardedModel
        from quantv2x_amd.engine import deploy
        # F batches in flight per rank, each with its own engine (workspace), graphs and stream; every rank issues the collectives in the
        # same round-robin order, so the F all-gathers in flight on one communicator cannot cross
        F = max(1, args.inflight)
        engines = [eng] + [deploy(state=state) for _ in range(F - 1)]
        if world == 1:
            # --force-sharded at world 1 times "the N>1 code path": the post stage must be the GENERAL a7-a11 kernels a real rank runs
            # (decode + warp + attention + GEMM heads), not the single-agent table look-up a world of one would take
            for e in engines:
                e.single_agent_tables = False
        streams = [torch.cuda.Stream() for _ in range(F)]
        pose = torch.from_numpy(poses[rank]).to(device)

        def make_step(ego_only):
            shardeds = [AgentShardedModel(e, frames=B, link=args.link, max_cav=wl["max_cav"], ego_only=ego_only, graph_link=args.graph_link)
                        for e in engines]
            for sh, st in zip(shardeds, streams):                     # capture (and the lazy one-off work) up front, slot by slot
                with torch.cuda.stream(st):
                    sh.forward(mine, pose)
                torch.cuda.synchronize()
            k = [0]

            def step():
                i = k[0] % F
                k[0] += 1
                with torch.cuda.stream(streams[i]):
                    shardeds[i].forward(mine, pose)
            return step, shardeds
        step, shardeds = make_step(args.ego_only)
        link_txt = "torch.distributed nccl = RCCL" if args.link == "torch" else "qv2x_allgather_codes (RCCL)"
        launch = (f"per rank: ONE hipGraph per step (a1-a6, {B} frames -> all-gather of code planes + poses [{link_txt}, captured] -> a7-a11)"
                  if args.graph_link else
                  f"per rank: hipGraph (a1-a6, {B} frames) -> all-gather of code planes + poses ({link_txt}) -> hipGraph (a7-a11)") + \
                 f"; {F} batches in flight on {F} streams"
        frames_per_step = B if args.ego_only else B * world
        print(f"[bench] rank {rank}/{world} on cuda:{local}: RCCL world size {dist.get_world_size()}", file=sys.stderr, flush=True)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, warmup, steps):
        for _ in range(warmup):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt = timed(step, args.warmup, args.steps)
    if sharded_mode and not args.ego_only:
        # the parity configuration of SURVEY 8(e)(i) beside the headline: only rank 0 fuses -- its output is the single-process model's
        del shardeds
        step_e, shardeds = make_step(True)
        ke = max(5, args.steps // 4)
        dte = timed(step_e, max(2, args.warmup // 4), ke)
        ego_line = {"value": round(B * ke / dte, 2), "unit": "frames/s", "steps": ke, "ms_per_step": round(dte / ke * 1e3, 4),
                    "frames_per_step": B, "note": "ego_only: every rank encodes and joins the all-gather, rank 0 alone runs a7-a11 (the reference's "
                    "single-ego output, SURVEY 8(e)(i)); `value` above counts every rank's own ego view (8(e)(ii))"}

    # p50 latency of ONE frame run alone (device-synchronised), outside the timed region
    lat = []
    if not sharded_mode:
        _, one, _, _ = frame_batch(1, 0, 1, device)
        solo = eng.capture(one)
    else:
        _, _, mine1, _ = frame_batch(world, rank, 1, device, layout=wl["layout"], max_cav=wl["max_cav"], own_only=world > 1)
        sh1 = AgentShardedModel(eng, frames=1, link=args.link, max_cav=wl["max_cav"], ego_only=args.ego_only, graph_link=args.graph_link)
        solo = lambda: sh1.forward(mine1, pose)
    for _ in range(10):
        solo()
    for _ in range(50):
        barrier()
        a = time.perf_counter()
        solo()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    lat.sort()
    barrier()
    ta = time.perf_counter()
    for _ in range(50):
        solo()
    barrier()
    one_at_a_time = 50 * (world if (sharded_mode and not args.ego_only) else 1) / (time.perf_counter() - ta)

    if rank == 0:
        hw = eng.fh * eng.fw
        line = {
            "metric": "frames/sec/node (N-agent int8 BEV fusion)", "value": round(frames_per_step * args.steps / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": wl["workload"], "baseline_config_index": wl["index"],
                       "stages": ("pfn+scatter, 19 conv + 3 deconv backbone, shrinker, 3-level codebook encode, " +
                                  ("every head by table look-up on the agent's own codes (single-agent scenes: AttFusion over one agent is the identity)"
                                   if world == 1 and not sharded_mode else "decode+warp+attention, heads (+ *_single heads)")),
                       "grid": wl["grid"], "agents_per_frame": world, "max_cav": wl["max_cav"], "layout": wl["layout"],
                       "heads": "multi-class (mc, 72 channels)" if wl["multiclass"] else "single-class (20 channels)",
                       "points_per_agent": N_POINTS,
                       "frames_per_step": frames_per_step, "batch_per_rank": B, "batches_in_flight": F,
                       "pillars_per_step_rank0": int((mine if sharded_mode else full["inputs_m1"])["voxel_features"].shape[0]),
                       "frame_definition": ("one ego-view fused detection frame; ego_only: rank 0 is the ego, the other ranks only encode and send" if args.ego_only
                                            else "one ego-view fused detection frame; with N GPUs every rank is the ego of its own view"),
                       "ego_only": bool(args.ego_only),
                       "launch": launch, "quantization": "W8A8 min-max PTQ (reference QuantModel recipe), random He-init weights",
                       "codebook_encode": (f"{eng.encode_mode}: exact integer candidates for every cell (int8 MFMA) + the reference-order fp32 chain on the cells a proven "
                                           "bound cannot decide -- the indices of the every-cell chain, by construction (quantv2x_amd/encode_two_stage.py)"
                                           if eng.encode_mode == "two_stage" else f"{eng.encode_mode}: every cell through the reference-order fp32 chain"),
                       "wire_bytes_per_agent_frame": 3 * hw,
                       "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1},
            "latency_ms_p50": round(lat[len(lat) // 2], 4), "latency_ms_p95": round(lat[int(len(lat) * 0.95) - 1], 4),
            "latency_note": "one frame run alone (batch 1, nothing else in flight), host-timed around launch + device sync",
            "value_one_frame_at_a_time": round(one_at_a_time, 2),
        }
        if ego_line is not None:
            line["ego_only"] = ego_line
        if sharded_mode and not args.no_extras:
            # north_star: ">= 6x 1 -> 8 at matched detection output".  The matched output is the ego-0 frame of every scene: this job's ego-only
            # figure over the SAME scenes (all `world` agents) on one GPU, measured here on rank 0 while the other ranks wait at the barrier below
            try:
                one = same_scene_one_gpu(eng, world, device, same_scene_scenes(world, B), iters=5)
                eo = (ego_line or {}).get("value") if not args.ego_only else line["value"]
                line["scaling_vs_one_gpu_same_scene"] = {"one_gpu": one, "n_gpu_ego_only_frames_per_s": eo,
                                                         "ratio": round(eo / one["fused_frames_per_s"], 3) if eo else None,
                                                         "note": "N-GPU ego-only fused frames/s (rank 0 fuses the all-gathered codes of the scene's agents) / the same scenes "
                                                                 "wholly on one GPU (every agent encoded here, ego = agent 0); both produce the reference's single-ego output"}
            except Exception as e:
                line["scaling_vs_one_gpu_same_scene"] = {"error": repr(e)[:300]}
        roof, stages = rooflines(eng, full if not sharded_mode else frame_batch(1, 0, B, device)[1], B, iters=max(10, args.steps // 5))
        line["roofline"], line["roofline_stages"] = roof, stages
        if world == 1 and not sharded_mode and not args.no_extras:
            # the same frame on the fp32 HIP path (the un-quantized model, engine_fp32.py): what W8A8 buys on this GPU
            from quantv2x_amd.engine import deploy as _deploy
            e32 = _deploy(fp_model)
            r32 = e32.capture(frame_batch(1, 0, 1, device)[1])
            for _ in range(3):
                r32()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(20):
                r32()
            torch.cuda.synchronize()
            ms32 = (time.perf_counter() - tb) / 20 * 1e3
            line["fp32_hip_path"] = {"ms_per_frame": round(ms32, 3), "frames_per_s": round(1e3 / ms32, 1),
                                     "note": "un-quantized model, one frame at a time, f32-MFMA convolutions; compare value_one_frame_at_a_time"}
            del e32, r32
            line["multi_agent_one_gpu"] = multi_agent_line(eng, device)
            line["pyramid_model"] = pyramid_model_line(device)
            line["second_encoder"] = second_encoder_line(device)
            line["encode_modes"] = encode_modes_line(state, full, B, F, args.steps, device)
            line["points_to_boxes"] = points_to_boxes_line(state, device)
            try:
                line["codebook_seg2_dict256"] = codebook_seg_line(full, B, device)
            except Exception as e:                                         # an extra must not cost the line
                line["codebook_seg2_dict256"] = {"error": repr(e)[:300]}
            # what a rank of the N-GPU lines executes per step, inside the driver's clock: worlds 2 and 4 reuse this
            # run's engine (same V2X-Real mc model); world 8 is BASELINE configs[4]'s OPV2V grid -- its model is calibrated here
            reh = {}
            for W in (2, 4, 8):
                try:
                    r = rehearse_line(W, B, 10, 2, device, engine=eng if W <= 4 else None)
                    reh[f"world{W}"] = {"ms_per_step": r["ms_per_step"], "frames_per_s_of_one_rank": r["value"], "stage_us": r["stage_us"],
                                        "workload": r["config"]["workload"], "pillars_per_step": r["config"]["pillars_per_step"],
                                        "frames_per_step": B}
                    # the LAST rank's step (its own sweeps, ego = W - 1: the every-rank-is-its-own-ego mode of the N-GPU line)
                    eng_w = r.pop("_engine")
                    rl = rehearse_line(W, B, 10, 2, device, engine=eng_w, rank=W - 1)
                    rl.pop("_engine")
                    if W == 8 and isinstance(line.get("multi_agent_one_gpu", {}).get("same_scene_b32"), dict):
                        try:                                                 # configs[4] wholly on one GPU: 4 scenes x 8 agents of the OPV2V grid per graph
                            line["multi_agent_one_gpu"]["same_scene_b32"]["8_agents_opv2v"] = same_scene_one_gpu(eng_w, 8, device, same_scene_scenes(8), iters=5)
                        except Exception as e:
                            line["multi_agent_one_gpu"]["same_scene_b32"]["8_agents_opv2v"] = {"error": repr(e)[:300]}
                    del eng_w
                    reh[f"world{W}"]["last_rank"] = {"rank": W - 1, "ego": W - 1, "ms_per_step": rl["ms_per_step"], "stage_us": rl["stage_us"],
                                                     "pillars_per_step": rl["config"]["pillars_per_step"],
                                                     "ms_per_step_over_rank0": round(rl["ms_per_step"] / r["ms_per_step"], 4)}
                except Exception as e:                                     # an extra must not cost the line
                    reh[f"world{W}"] = {"error": repr(e)[:300]}
            reh["note"] = ("rank 0's step of `bench.py --gpus W` rehearsed on ONE GPU (AgentShardedModel(emulate_world=W)): a1-a6 on the own agent's "
                           "frames, the own payload copied into every agent slot where the all-gather would be, then pairwise + a7-a11 over W agents. "
                           "NOT a multi-GPU measurement: no link, and the other agents are copies of the own code planes")
            line["rehearsal"] = reh
            try:
                line["sharded_world1"] = sharded_world1_line(state, B, device)
            except Exception as e:
                line["sharded_world1"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1 and not sharded_mode:       # reported on rank 0 at N = 1 only
            # the CPU checker on frames of THIS batch; each frame it finishes is compared with what the timed graphs computed (both
            # engines' workspaces hold their last replay: same inputs, so the same bytes) -- the parity gate of SURVEY 8(d)
            from quantv2x_amd import synth
            for i in range(F):                                                  # one more replay each: the extras above reused engine 0's workspace
                with torch.cuda.stream(streams[i]):
                    reps[i]()
            torch.cuda.synchronize()
            scenes_np = [synth.make_scene(SHAPE, n_agents=1, seed=3 + f, n_points=N_POINTS, layout=wl["layout"], max_cav=wl["max_cav"]) for f in range(min(B, 12))]
            wss = [e._workspace(B) for e in engines]

            def gpu_check(f, taps):
                import numpy as np
                want_codes = taps["codes"].reshape(taps["codes"].shape[0], -1)                       # [levels, H*W] of one agent
                want_u8 = taps["shrinker_m1.layers.0.double_conv.1"][0]                             # [H, W, 256] uint8
                im = um = 0
                for ws in wss:
                    im += int((ws["codes"][:, f].cpu().numpy() != want_codes).sum())
                    got = (ws["s1"][f, 1:-1, 1:-1, :].to(torch.int16) + 128).to(torch.uint8).cpu().numpy()
                    um += int((got != want_u8).sum())
                return im, um
            line["cpu_baseline"], parity = cpu_baseline(state, scenes_np, gpu_check)
            parity.update({"engines_checked": F, "indices_per_frame": 3 * hw, "u8_per_frame": hw * 256,
                           "what": "codebook indices (3 levels) and the shrinker's uint8 output map of the batch's first frames, as left in the workspaces of "
                                   "the timed HIP graphs, against oracle/ on the same frames; must be 0 / 0"})
            line["parity"] = parity
            line["cpu_baseline_torch"] = cpu_baseline_torch(fp_model, qt, scenes_np[0], cores)
    else:
        line = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()     # RCCL prints its version banner around here: keep the JSON line last
    import ctypes
    ctypes.CDLL(None).fflush(None)       # RCCL's version banner sits in the C stdio buffer: push it out (to stderr, see above) first
    sys.stdout.flush()
    if saved_stdout is not None:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
