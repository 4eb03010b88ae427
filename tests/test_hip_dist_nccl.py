"""The multi-GPU driver (quantv2x_amd/dist.py) on the REAL engine over RCCL (backend "nccl").

world_size 1 runs inside this process (the collective degenerates to a copy / a one-rank ncclAllGather, everything else --
payload layout, pose block, pairwise kernel, the two HIP graphs around the collective, the batched frames -- is the N-GPU
code path); the 2-process variant starts fresh children before any GPU call and is skipped on a one-GPU box."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from _common import calibrated_plugin, scene_np

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_pairwise_from_poses_kernel_is_the_oracle_bit_for_bit():
    from oracle import geometry
    from quantv2x_amd import lib as L, synth
    lib = L.load()
    for layout, n, cav in (("ring", 8, 8), ("line", 3, 5), ("ring", 1, 5)):
        poses = synth.agent_poses(n, layout)
        stride, off = 256, 64
        g = np.zeros((n, stride), np.uint8)
        for a, p in enumerate(poses):
            g[a, off:off + 128] = np.ascontiguousarray(p, dtype=np.float64).view(np.uint8).reshape(-1)
        gt = torch.from_numpy(g).cuda()
        out = torch.full((cav, cav, 4, 4), -7.0, dtype=torch.float64, device="cuda")
        L.check(lib.qv2x_pairwise_from_poses_f64(L.ptr(gt), n, stride, off, cav, L.ptr(out), L.current_stream()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), geometry.pairwise_from_poses(poses, cav))
        np.testing.assert_allclose(out.cpu().numpy(), synth.pairwise_t_matrix(poses, cav), rtol=0, atol=1e-12)   # np.linalg.solve
    assert lib.qv2x_pairwise_from_poses_f64(L.ptr(gt), 2, 100, 64, 5, L.ptr(out), None) == -2      # pose block outside the stride


@pytest.fixture(scope="module")
def nccl_world1():
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("link,frames,graphs", [("torch", 1, True), ("torch", 2, True), ("rccl", 2, True), ("rccl", 1, False)])
def test_sharded_driver_world1_equals_forward(nccl_world1, link, frames, graphs):
    from quantv2x_amd import synth
    from quantv2x_amd.dist import AgentShardedModel, payload_layout
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin())
    eng = deploy(state=state)
    # `frames` single-agent frames of this rank's agent in one step
    parts = []
    for f in range(frames):
        sc = scene_np(1, seed=3 + f)
        part = {k: v.copy() for k, v in sc["inputs_m1"].items()}
        part["voxel_coords"][:, 0] = f
        parts.append(part)
    inp = {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).cuda() for k in parts[0]}
    pose = torch.from_numpy(synth.pose_matrix(3.0, -1.0, 0.2)).cuda()
    model = AgentShardedModel(eng, frames=frames, link=link, graphs=graphs, max_cav=5)
    first = {k: v.clone() for k, v in model.forward(inp, pose).items()}
    again = model.forward(inp, pose)                                   # graph replay (or the second eager step)
    torch.cuda.synchronize()
    assert (model._captured is not None) == graphs
    cbytes, pose_off, pbytes = payload_layout(3, frames, eng.fh * eng.fw)
    assert tuple(model.gathered.shape) == (1, pbytes)
    got_pose = model.gathered[0, pose_off:pose_off + 128].cpu().numpy().view(np.float64).reshape(4, 4)
    np.testing.assert_array_equal(got_pose, synth.pose_matrix(3.0, -1.0, 0.2))
    # the same frames through the single-process forward: a batch of `frames` scenes with one agent each
    dd = {"inputs_m1": inp, "agent_modality_list": ["m1"] * frames, "record_len": torch.ones(frames, dtype=torch.int64),
          "pairwise_t_matrix": torch.eye(4, dtype=torch.float64).expand(frames, 5, 5, 4, 4).contiguous().cuda()}
    want = eng(dd)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(first[k], want[k]), k
        assert torch.equal(again[k], want[k]), k
    model.close()


def test_capture_is_rank_local_and_a_new_pose_does_not_recapture(nccl_world1):
    """One collective per step, none from capturing: a rank that re-captures alone (its input buffers moved) must not issue an
    all-gather the other ranks never see -- every gather has the same size, so it would pair with their next step.  A fresh pose
    tensor per step (what a data loader hands over) re-captures nothing."""
    from quantv2x_amd import synth
    from quantv2x_amd.dist import AgentShardedModel
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    eng = deploy(state=export_ptq_state(calibrated_plugin()))
    inp = {k: torch.from_numpy(v.copy()).cuda() for k, v in scene_np(1)["inputs_m1"].items()}
    model = AgentShardedModel(eng, frames=1, link="torch", graphs=True, max_cav=5)
    calls = [0]
    inner = model._exchange

    def counted():
        calls[0] += 1
        inner()
    model._exchange = counted
    p0 = torch.from_numpy(synth.pose_matrix(3.0, -1.0, 0.2)).cuda()
    a = model.forward(inp, p0)["preds_tensor"].clone()
    assert calls[0] == 1
    graphs = model._captured[1:3]
    p1 = torch.from_numpy(synth.pose_matrix(3.0, -1.0, 0.2)).cuda()           # same pose, another tensor
    b = model.forward(inp, p1)["preds_tensor"].clone()
    assert calls[0] == 2 and model._captured[1:3] == graphs
    moved = {k: v.clone() for k, v in inp.items()}                           # the input buffers moved: this rank re-captures on its own
    c = model.forward(moved, p1)["preds_tensor"].clone()
    torch.cuda.synchronize()
    assert calls[0] == 3 and model._captured[1] is not graphs[0]
    assert torch.equal(a, b) and torch.equal(a, c)
    model.close()


@pytest.mark.parametrize("link,frames", [("torch", 1), ("rccl", 2)])
def test_sharded_driver_world1_pyramid_model(nccl_world1, link, frames):
    """The same driver over the Pyramid engine: the payload is the 64-wide codebook's planes, the ego side is decode_features."""
    from _common import calibrated_pyramid_plugin
    from quantv2x_amd import synth
    from quantv2x_amd.dist import AgentShardedModel
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    eng = deploy(state=export_ptq_state(calibrated_pyramid_plugin()))
    parts = []
    for f in range(frames):
        part = {k: v.copy() for k, v in scene_np(1, seed=3 + f)["inputs_m1"].items()}
        part["voxel_coords"][:, 0] = f
        parts.append(part)
    inp = {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).cuda() for k in parts[0]}
    pose = torch.from_numpy(synth.pose_matrix(3.0, -1.0, 0.2)).cuda()
    model = AgentShardedModel(eng, frames=frames, link=link, graphs=True, max_cav=5)
    first = model.forward(inp, pose)["preds_tensor"].clone()
    again = model.forward(inp, pose)["preds_tensor"]
    dd = {"inputs_m1": inp, "agent_modality_list": ["m1"] * frames, "record_len": torch.ones(frames, dtype=torch.int64),
          "pairwise_t_matrix": torch.eye(4, dtype=torch.float64).expand(frames, 5, 5, 4, 4).contiguous().cuda()}
    want = eng(dd)["preds_tensor"]
    torch.cuda.synchronize()
    assert torch.equal(first, want) and torch.equal(again, want)
    model.close()


def test_two_processes_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: the 2-rank RCCL exchange needs two")
    for link in ("torch", "rccl"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_nccl_worker.py"), link]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "OK sharded == single-process" in r.stdout and "RCCL world size 2" in r.stdout
