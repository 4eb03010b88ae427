#!/usr/bin/env python
"""bench.py -- frames/s of the quantized per-agent encode + intermediate-fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the whole hot path (a1-a11: PFN + scatter, int8 backbone + shrinker, codebook encode,
exchange, decode + warp + attention, heads) over one synthetic V2X-Real-shaped frame.  With N GPUs, rank r owns
agent r (one process per GPU, RCCL all-gather of the code planes) and every rank is the ego of its own view, so a
step produces N fused N-agent frames; ``value`` = N * K / (max-over-ranks time).  Per-GPU work is fixed as N grows
(one agent encoded per GPU): ``"scaling": "weak"``.

Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task prompt),
with ``roofline`` for the dominant kernel (the wide-layer int8 MFMA convolution) measured live with HIP events,
and ``cpu_baseline`` = the CPU oracle (``oracle/``, the checker -- never the product) timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_MFMA_PEAK_TOPS = 5000.0     # dense int8 peak of MI355X (2 x the 2.5 PF bf16 dense peak), MI355X_MICROARCH.md
SHAPE = "v2xreal"
N_POINTS = 60000


def build_engine(n_threads):
    import copy
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(n_threads)
    # the reference's flow: create_model -> load weights (seeded synthetic: no checkpoint exists here) -> QuantModel
    # -> weight quantizers -> one min-max observer pass (torch, on the host) -> freeze -> deploy on the HIP path
    model = train_utils.create_model(copy.deepcopy(synth.make_hypes(SHAPE))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    calib = synth.scene_to_torch(synth.make_scene(SHAPE, n_agents=1, seed=3, n_points=N_POINTS))
    qt = inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib])
    state = export_ptq_state(qt)
    return state, deploy(state=state)


def my_scene(world, rank, device):
    """Scene with `world` agents; returns (inputs of agent `rank` re-indexed to batch 0, full scene on device)."""
    import torch
    from quantv2x_amd import synth
    sc = synth.make_scene(SHAPE, n_agents=world, seed=3, n_points=N_POINTS, layout="ring" if world > 2 else "line")
    full = synth.scene_to_torch(sc, device)
    co = full["inputs_m1"]["voxel_coords"]
    mine = co[:, 0] == rank
    inp = {"voxel_features": full["inputs_m1"]["voxel_features"][mine].contiguous(),
           "voxel_coords": co[mine].clone().contiguous(),
           "voxel_num_points": full["inputs_m1"]["voxel_num_points"][mine].contiguous()}
    inp["voxel_coords"][:, 0] = 0
    return sc, full, inp


def conv_roofline(eng, iters):
    """Average launch duration and algorithmic ops of the dominant kernel: the wide-layer int8 MFMA convolution
    ``conv3x3_i8_wide_kernel<5, true, 8, 1>`` -- one launch per frame (the shrinker's 3x3 384 -> 256 convolution over the
    three-scale concat, 31.14 GMAC = 40 % of all conv work), timed with HIP events on the launch stream."""
    import torch
    pick = lambda kind, layer: kind == "conv" and layer is eng.shrink0
    launches = [p for p in eng.conv_plan(1) if pick(p[0], p[1])]
    ops = sum(2.0 * p[7] for p in launches)
    for _ in range(3):
        eng.run_plan(1, only=pick)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        eng.run_plan(1, only=pick)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (iters * len(launches))
    achieved = ops / len(launches) / (us * 1e-6) / 1e12
    traffic = None       # HBM-side bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (not live)
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_wide_conv.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch")
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
            "frac": round(achieved / INT8_MFMA_PEAK_TOPS, 4), "traffic": traffic,
            "traffic_note": "bytes per launch, FETCH_SIZE + WRITE_SIZE of profiles/r01_pmc_wide_conv.json (algorithmic: 23.4 MB)",
            "kernel": "conv3x3_i8_wide_kernel<5, true, 8, 1>", "launches_per_frame": len(launches),
            "avg_launch_us": round(us, 2), "algorithmic_gop_per_launch": round(ops / len(launches) / 1e9, 3)}


def stage_times(eng, dd, iters=20):
    """Per-stage ms of one eager frame (outside the timed region)."""
    import torch
    n = len(dd["agent_modality_list"])
    hw = eng.fh * eng.fw
    stages = {}

    def timed(name, fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        stages[name] = round(e0.elapsed_time(e1) / iters, 4)
    timed("pfn_scatter", lambda: eng.pillars_to_canvas(dd["inputs_m1"], n))
    timed("backbone_convs", lambda: eng.run_plan(n, only=lambda k, l: k == "conv" and l.name.startswith("backbone")))
    timed("backbone_deconvs", lambda: eng.run_plan(n, only=lambda k, l: k == "deconv"))
    timed("shrinker_convs", lambda: eng.run_plan(n, only=lambda k, l: k == "conv" and l.name.startswith("shrinker")))
    timed("codebook_encode", lambda: eng.encode_codes(n))
    codes = eng._workspace(n)["codes"]
    pw = dd["pairwise_t_matrix"][0].contiguous()
    timed("fuse_and_heads", lambda: eng.fuse_and_heads(codes, hw, n * hw, pw, n, 0))
    return stages


def frames_in_flight(state, dd, steps, n_flight=2):
    """Throughput with ``n_flight`` independent frames in flight (one engine + HIP graph + stream each): the small backbone
    layers leave most CUs idle, a second frame fills them.  Reported next to ``value`` (one frame at a time), SURVEY.md
    §8(d) "throughput also at the best frame-batch"."""
    import torch
    from quantv2x_amd.engine import deploy
    engs = [deploy(state=state) for _ in range(n_flight)]
    streams = [torch.cuda.Stream() for _ in range(n_flight)]
    reps = []
    for e, st in zip(engs, streams):
        with torch.cuda.stream(st):
            reps.append(e.capture(dd))
    torch.cuda.synchronize()

    def round_():
        for r, st in zip(reps, streams):
            with torch.cuda.stream(st):
                r()
    for _ in range(10):
        round_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        round_()
    torch.cuda.synchronize()
    return round(n_flight * steps / (time.perf_counter() - t0), 2)


def cpu_baseline(state, sc_np, budget_s=12.0, max_frames=6):
    """The CPU oracle (checker) on the same workload, on the host cores of this box."""
    from oracle.spec import Oracle
    cores = os.cpu_count() or 1
    orc = Oracle(state)
    orc.forward(sc_np)                      # warm-up (also builds the shared object if needed)
    t0, frames = time.time(), 0
    while frames < max_frames and (time.time() - t0) < budget_s:
        orc.forward(sc_np)
        frames += 1
    dt = time.time() - t0
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{frames} full single-agent V2X-Real frames (whole hot path) through oracle/ in {dt:.1f} s, OpenMP on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inflight", action="store_true", help="skip the extra 2-frames-in-flight throughput measurement")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (eager launches + all_gather of the code planes) even with one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    cores = os.cpu_count() or 8
    state, eng = build_engine(max(1, min(32, cores // max(world, 1))))
    sc_np, full, mine = my_scene(world, rank, device)
    pairwise = full["pairwise_t_matrix"][0].contiguous()

    if world == 1 and not args.force_sharded:
        step = eng.capture(full)            # HIP graph of the whole frame
        launch = "hipGraph replay of the whole frame"
    else:
        from quantv2x_amd.dist import AgentShardedModel
        sharded = AgentShardedModel(eng)
        step = lambda: sharded.forward(mine, pairwise)
        launch = "eager launches + RCCL all_gather_into_tensor of the code planes"

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # p50 per-frame latency (one frame at a time, device-synchronised), outside the timed region
    lat = []
    for _ in range(min(50, args.steps)):
        torch.cuda.synchronize(); a = time.perf_counter()
        step(); torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    lat.sort()

    if rank == 0:
        line = {
            "metric": "frames/sec/node (N-agent int8 BEV fusion)", "value": round(world * args.steps / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": "Single-agent int8 PointPillar + BEV backbone on 1xMI355X, synthetic V2X-Real point cloud "
                                   "(~60k pts, 0.4 m voxels)" if world == 1 else
                                   f"{world}-agent intermediate fusion with codebook-compressed BEV features, one agent per GPU",
                       "stages": "pfn+scatter, 19 conv + 3 deconv backbone, shrinker, 3-level codebook encode, decode+warp+attention, heads (+ *_single heads)",
                       "grid": "704x200x1 voxels -> 256x100x352 feature map", "agents_per_frame": world,
                       "pillars_agent0": int(mine["voxel_features"].shape[0]) if world > 1 else int(full["inputs_m1"]["voxel_features"].shape[0]),
                       "frame_definition": "one ego-view fused detection frame; with N GPUs every rank is the ego of its own view",
                       "launch": launch, "quantization": "W8A8 min-max PTQ (reference QuantModel recipe), random He-init weights"},
            "latency_ms_p50": round(lat[len(lat) // 2], 4), "latency_ms_p95": round(lat[int(len(lat) * 0.95) - 1], 4),
        }
        line["roofline"] = conv_roofline(eng, iters=max(10, args.steps // 4))
        solo = full if world == 1 else None
        if solo is not None:
            line["stage_ms"] = stage_times(eng, solo)
            if not args.no_inflight:
                line["throughput_2_frames_in_flight"] = frames_in_flight(state, solo, max(20, args.steps // 2))
        if not args.no_cpu_baseline and world == 1:       # reported on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(state, sc_np)
    else:
        line = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()     # RCCL prints its version banner around here: keep the JSON line last
    if line is not None:
        import ctypes
        ctypes.CDLL(None).fflush(None)   # RCCL's version banner sits in the C stdio buffer: push it out before the JSON line
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
