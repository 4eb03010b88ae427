"""The agent-sharded (one agent per rank) path on CPU with the gloo backend, world_size 2 and 3.

The HIP encode cannot run here, so a stand-in engine produces each rank's code planes with the CPU oracle; what is
under test is the exchange: layout of the gathered tensor, the strides handed to the fuse kernel, the ego index,
and that fusing the gathered codes equals the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _common import calibrated_plugin, ego_view, scene_np


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleEngine:
    """The stage interface of DeployedModel that AgentShardedModel drives, backed by the CPU oracle (tests only)."""
    has_codebook = True

    def __init__(self, state):
        from oracle.spec import Oracle
        self.orc = Oracle(state)
        self.calls = []
        nx, ny = self.orc.nx, self.orc.ny
        self.h, self.w = ny // 2, nx // 2

    def wire_shape(self):
        return 3, self.h * self.w

    def encode_into(self, inputs, frames, codes_out):
        sc = {"inputs_m1": {k: v.numpy() for k, v in inputs.items()}}
        _, canvas, cq = self.orc.pfn_scatter(sc, frames)
        cat, cat_q = self.orc.backbone(canvas, cq)
        shr, shr_q = self.orc.shrinker(cat, cat_q)
        codes = self.orc.encode(shr, shr_q)                       # [levels, frames*hw]
        codes_out.copy_(torch.from_numpy(codes.reshape(codes.shape[0], frames, -1)))

    def pairwise_from_poses(self, gathered, world, agent_stride, pose_offset, max_cav, out):
        from oracle import geometry
        g = gathered.numpy().reshape(world, -1)
        poses = [g[a, pose_offset:pose_offset + 128].copy().view(np.float64).reshape(4, 4) for a in range(world)]
        out.copy_(torch.from_numpy(geometry.pairwise_from_poses(poses, max_cav)))

    def fuse_frames_and_heads(self, gathered, agent_stride, level_stride, frame_stride, pairwise, n_agents, ego, own_codes, frames):
        self.calls.append((tuple(gathered.shape), agent_stride, level_stride, frame_stride, n_agents, ego))
        flat = gathered.numpy().reshape(-1)
        hw, outs = self.h * self.w, []
        for f in range(frames):
            planes = np.stack([[flat[a * agent_stride + l * level_stride + f * frame_stride:][:hw] for a in range(n_agents)] for l in range(3)])
            feats = self.orc.decode(planes.reshape(3, -1)).reshape(n_agents, self.h, self.w, 256)
            order = [ego] + [a for a in range(n_agents) if a != ego]
            # the oracle fuses in agent order with agent 0 as the ego: present the ego first and its pairwise row
            t = pairwise[f].numpy()[None]
            t_ego = t[:, order][:, :, order]
            fused = self.orc.fuse(feats[order], t_ego, [n_agents])
            outs.append(np.concatenate(self.orc.heads(fused), axis=1))
        return {"preds_tensor": torch.from_numpy(np.concatenate(outs))}


def _worker(rank, world, port, state, out_q, frames):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantv2x_amd import synth
        from quantv2x_amd.dist import AgentShardedModel
        torch.set_num_threads(1)
        # `frames` scenes (different sweeps, same poses); this rank contributes agent `rank` of each, batch index = frame
        parts = []
        for f in range(frames):
            sc = scene_np(world, seed=3 + f)
            co = sc["inputs_m1"]["voxel_coords"]
            mine = co[:, 0] == rank
            part = {k: v[mine].copy() for k, v in sc["inputs_m1"].items()}
            part["voxel_coords"][:, 0] = f
            parts.append(part)
        inp = {k: torch.from_numpy(np.concatenate([p[k] for p in parts])) for k in parts[0]}
        pose = torch.from_numpy(synth.agent_poses(world, "line")[rank])
        eng = _OracleEngine(state)
        model = AgentShardedModel(eng, frames=frames)
        out = model.forward(inp, pose)
        calls = list(eng.calls)
        ego_only = AgentShardedModel(eng, ego_only=True, frames=frames).forward(inp, pose)
        assert (ego_only is None) == (rank != 0)
        assert len(eng.calls) == len(calls) + (1 if rank == 0 else 0)          # ego_only: rank 0 alone runs the post stage
        out_q.put((rank, out["preds_tensor"].numpy(), model.gathered.numpy().copy(), calls, model.pairwise.numpy().copy(),
                   None if ego_only is None else ego_only["preds_tensor"].numpy()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,frames", [(2, 1), (3, 1), (2, 2), (4, 1), (8, 1)])
def test_sharded_agents_match_single_process(world, frames):
    """worlds 2, 3, 4 (BASELINE configs[3]) and 8 (configs[4]): every rank the ego of its own view AND ``ego_only`` (rank 0 alone fuses: the
    reference's single-ego output) against the single-process model at tiny shape"""
    from quantv2x_amd import synth
    from quantv2x_amd.dist import payload_layout
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle import geometry
    from oracle.spec import Oracle
    state = export_ptq_state(calibrated_plugin())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, state, q, frames)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    orc = Oracle(state)
    wants, codes = [], []
    for f in range(frames):
        taps = {}
        sc = scene_np(world, seed=3 + f)
        wants.append(orc.forward(sc, taps)["preds_tensor"])
        codes.append(taps["codes"].reshape(3, world, -1))
    hw = codes[0].shape[-1]
    cbytes, pose_off, pbytes = payload_layout(3, frames, hw)
    poses = synth.agent_poses(world, "line")
    for rank, preds, gathered, calls, pairwise, ego_preds in results:
        assert gathered.shape == (world, pbytes)
        for a in range(world):                                                   # agent-major wire layout: codes, then the pose
            planes = gathered[a, :cbytes].reshape(3, frames, hw)
            for f in range(frames):
                np.testing.assert_array_equal(planes[:, f], codes[f][:, a])
            for f in range(frames):
                got_pose = gathered[a, pose_off + 128 * f: pose_off + 128 * (f + 1)].copy().view(np.float64).reshape(4, 4)
                np.testing.assert_array_equal(got_pose, poses[a])
        shape, a_stride, l_stride, f_stride, n, ego = calls[0]
        assert (a_stride, l_stride, f_stride, n, ego) == (pbytes, frames * hw, hw, world, rank)
        np.testing.assert_array_equal(pairwise[0], geometry.pairwise_from_poses(poses, world))
        # the pairwise matrix built from the gathered poses is the dataset's (np.linalg.solve) up to rounding
        np.testing.assert_allclose(pairwise[0], scene_np(world)["pairwise_t_matrix"][0][:world, :world], rtol=0, atol=1e-12)
    want = np.concatenate(wants)
    np.testing.assert_allclose(results[0][1], want, rtol=1e-5, atol=1e-6)   # rank 0 = the reference's ego
    np.testing.assert_array_equal(results[0][5], results[0][1])             # ego_only: the same output on rank 0 ...
    assert all(r[5] is None for r in results[1:])                            # ... and none on the others
    # Every other rank is the ego of its own view (ego = rank): the single-process oracle on the scene presented ego-first -- agents
    # [r] + others, the DATASET's pairwise matrix (np.linalg.solve on the agents' poses in that order, transformation_utils.py:21-66)
    # permuted the same way; the reference keeps agent 0's row of that scene (fusion_in_one.py:131-151).  What is under test is the driver:
    # the ego index it hands on, its pairwise matrices from the gathered poses, the wire layout.
    h, w = orc.ny // 2, orc.nx // 2
    for rank, preds, _, _, _, _ in results[1:]:
        assert preds.shape == want.shape
        for f in range(frames):
            sc = scene_np(world, seed=3 + f)
            feats = orc.decode(np.ascontiguousarray(codes[f]).reshape(3, -1)).reshape(world, h, w, 256)
            f2, t2 = ego_view(feats, sc["pairwise_t_matrix"][0], world, rank)
            view = np.concatenate(orc.heads(orc.fuse(f2, t2, [world])), axis=1)
            np.testing.assert_allclose(preds[f:f + 1], view, rtol=1e-5, atol=1e-6, err_msg=f"rank {rank} as the ego, frame {f}")
            assert not np.allclose(view, wants[f])                                # (and that view is not rank 0's)


def test_exchange_world_one_is_a_copy():
    from quantv2x_amd.dist import exchange_codes, gathered_strides
    port = _free_port()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        c = torch.arange(3 * 8, dtype=torch.uint8).view(3, 8)
        g = exchange_codes(c)
        assert g.shape == (1, 3, 8) and torch.equal(g[0], c)
        assert gathered_strides(3, 8) == (24, 8)
        from quantv2x_amd.dist import payload_layout
        assert payload_layout(3, 1, 35200) == (105600, 105600, 105728) and payload_layout(3, 2, 15) == (90, 96, 352)
    finally:
        dist.destroy_process_group()


def test_rehearsal_of_a_world_of_three_on_one_process():
    """``emulate_world`` (bench.py --rehearse-world): one process plays rank 0 of a world of W -- every agent slot holds the own code planes,
    the other agents' poses come from the caller -- so the post stage sees W agents, the ego at slot 0, and the pairwise matrices of the
    given poses.  No process group is needed."""
    from quantv2x_amd import synth
    from quantv2x_amd.dist import AgentShardedModel, POSE_BYTES
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle import geometry
    state = export_ptq_state(calibrated_plugin())
    eng = _OracleEngine(state)
    W = 3
    poses = torch.from_numpy(np.stack(synth.agent_poses(W, "ring")))
    sc = scene_np(1)
    mine = {k: torch.from_numpy(v) for k, v in sc["inputs_m1"].items()}
    sh = AgentShardedModel(eng, frames=1, max_cav=5, emulate_world=W, emulate_poses=poses)
    out = sh.forward(mine, poses[0])
    assert sh.world == W and sh.gathered.shape[0] == W
    shape, agent_stride, level_stride, frame_stride, n_agents, ego = eng.calls[-1]
    assert (n_agents, ego) == (W, 0) and agent_stride == sh.payload_bytes
    for a in range(W):                                              # the own code planes in every slot, that agent's pose beside them
        assert torch.equal(sh.gathered[a, :sh.codes_bytes], sh.payload[:sh.codes_bytes])
        got = sh.gathered[a, sh.pose_off:sh.pose_off + POSE_BYTES].numpy().view(np.float64).reshape(4, 4)
        np.testing.assert_array_equal(got, poses[a].numpy())
    np.testing.assert_array_equal(sh.pairwise[0].numpy(), geometry.pairwise_from_poses([p.numpy() for p in poses], 5))
    assert np.isfinite(out["preds_tensor"].numpy()).all()
    with pytest.raises(ValueError):
        AgentShardedModel(eng, emulate_world=W, emulate_poses=poses[:2])
    # --rehearse-rank r: the same world played as rank r -- ego = r in the post stage (the every-rank-is-its-own-ego mode); with
    # ``ego_only`` a rank r > 0 runs no post stage at all
    for r in (1, 2):
        shr = AgentShardedModel(eng, frames=1, max_cav=5, emulate_world=W, emulate_poses=poses, emulate_rank=r)
        out_r = shr.forward(mine, poses[r])
        assert eng.calls[-1][4:] == (W, r) and shr.rank == r
        np.testing.assert_array_equal(shr.pairwise[0].numpy(), sh.pairwise[0].numpy())
        assert out_r["preds_tensor"].shape == out["preds_tensor"].shape
    n_calls = len(eng.calls)
    assert AgentShardedModel(eng, frames=1, max_cav=5, emulate_world=W, emulate_poses=poses, emulate_rank=2, ego_only=True).forward(mine, poses[2]) is None
    assert len(eng.calls) == n_calls
    with pytest.raises(ValueError):
        AgentShardedModel(eng, emulate_world=W, emulate_poses=poses, emulate_rank=W)
    with pytest.raises(ValueError):
        AgentShardedModel(eng, emulate_rank=1)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks through torch.distributed.run (a fresh child) and
    relays rank 0's line; the dry-run flag keeps the ranks on gloo / CPU.  Without two GPUs the real bench refuses instead of
    printing an `n_gpus: 1` line for a job that did not run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-ranks"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_joined"] and line["steps"] == 3
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "n_gpus" not in r.stdout
    # a launcher whose world size contradicts --gpus is an error, not a silently different job
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run-ranks"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_workload_follows_baseline_configs_per_world_size():
    """`bench.py --gpus N` runs the configuration BASELINE.json names for N: 1 -> configs[1], 2 -> configs[2] (V2X-Real, line), 4 ->
    configs[3] (V2X-Real VC, ring; max_cav 5, mc heads: lidar_attfuse_stage3.yaml:12), 8 -> configs[4] (OPV2V 512 x 512 grid,
    opv2v/LiDAROnly/lidar_attfuse.yaml:17, max_cav 8, single-class heads, >= 40k pillars per agent).  The table itself, the synthetic
    sweep it promises, and the launcher path at world 8 (gloo, no GPU) and at world 1 (used to die in init_process_group)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from quantv2x_amd import synth
    want = {1: (1, "v2xreal", True, 5, "line"), 2: (2, "v2xreal", True, 5, "line"), 3: (3, "v2xreal", True, 5, "ring"), 4: (3, "v2xreal", True, 5, "ring"),
            8: (4, "opv2v", False, 8, "ring")}
    for n, (idx, shape, mc, cav, layout) in want.items():
        wl = bench.workload_for(n)
        assert (wl["index"], wl["shape"], wl["multiclass"], wl["max_cav"], wl["layout"]) == (idx, shape, mc, cav, layout), (n, wl)
    # the one-GPU same-scene denominators of the scaling figure: 32 scenes per graph on V2X-Real, 4 x 8 agents on OPV2V
    assert [bench.same_scene_scenes(n) for n in (2, 3, 4, 8)] == [32, 32, 32, 4] and callable(bench.same_scene_one_gpu)
    w8 = bench.workload_for(8)
    assert "8-agent OPV2V-H" in w8["workload"] and "8xMI355X" in w8["workload"]
    sc = synth.make_scene(w8["shape"], n_agents=1, seed=3, n_points=w8["n_points"], max_cav=w8["max_cav"])
    assert sc["inputs_m1"]["voxel_features"].shape[0] >= 40000 and sc["pairwise_t_matrix"].shape[1] == 8
    hy = synth.make_hypes(w8["shape"], multiclass=False)
    assert synth.grid_size(*synth.SHAPES["opv2v"][:2])[:2] == (512, 512)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run-ranks", "--ego-only"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    c = line["config"]
    assert line["n_gpus"] == 8 and line["ranks_joined"] and line["ego_only"]
    assert c["baseline_config_index"] == 4 and c["shape"] == "opv2v" and c["voxel_grid"] == [512, 512] and c["feature_map"] == [256, 256]
    assert c["max_cav"] == 8 and c["multiclass_heads"] is False and c["agents_per_frame"] == 8 and "OPV2V-H" in c["workload"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run-ranks"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["baseline_config_index"] == 1 and line["config"]["shape"] == "v2xreal"
