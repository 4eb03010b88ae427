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

from _common import calibrated_plugin, scene_np


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleEngine:
    """Same three stage methods as DeployedModel, backed by the CPU oracle (tests only)."""

    def __init__(self, state):
        from oracle.spec import Oracle
        self.orc = Oracle(state)
        self.calls = []

    def encode_agents(self, inputs, n_agents):
        sc = {"inputs_m1": {k: v.numpy() for k, v in inputs.items()}}
        _, canvas, cq = self.orc.pfn_scatter(sc, n_agents)
        cat, cat_q = self.orc.backbone(canvas, cq)
        shr, shr_q = self.orc.shrinker(cat, cat_q)
        self.hw = shr.shape[1] * shr.shape[2]
        self.shape = shr.shape
        codes = self.orc.encode(shr, shr_q)                       # [levels, n*hw]
        return torch.from_numpy(codes.reshape(codes.shape[0], n_agents, -1))

    def fuse_and_heads(self, codes, agent_stride, level_stride, pairwise_b, n_agents, ego=0):
        self.calls.append((tuple(codes.shape), agent_stride, level_stride, n_agents, ego))
        flat = codes.numpy().reshape(-1)
        levels = codes.shape[1]
        planes = np.stack([[flat[a * agent_stride + l * level_stride: a * agent_stride + l * level_stride + self.hw]
                            for a in range(n_agents)] for l in range(levels)])            # [levels, A, hw]
        feats = self.orc.decode(planes.reshape(levels, -1)).reshape(n_agents, self.shape[1], self.shape[2], 256)
        order = [ego] + [a for a in range(n_agents) if a != ego]
        # the oracle fuses in agent order with agent 0 as the ego: present the ego first and its pairwise row
        t = pairwise_b.numpy()[None]
        t_ego = t[:, order][:, :, order]
        fused = self.orc.fuse(feats[order], t_ego, [n_agents])
        return {"preds_tensor": torch.from_numpy(np.concatenate(self.orc.heads(fused), axis=1))}


def _worker(rank, world, port, state, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantv2x_amd.dist import AgentShardedModel
        torch.set_num_threads(1)
        sc = scene_np(world)
        co = sc["inputs_m1"]["voxel_coords"]
        mine = co[:, 0] == rank
        inp = {k: torch.from_numpy(v[mine].copy()) for k, v in sc["inputs_m1"].items()}
        inp["voxel_coords"][:, 0] = 0
        eng = _OracleEngine(state)
        model = AgentShardedModel(eng)
        pw = torch.from_numpy(sc["pairwise_t_matrix"][0])
        out = model.forward(inp, pw)
        out_q.put((rank, out["preds_tensor"].numpy(), model._gathered.numpy().copy(), eng.calls))
        ego_only = AgentShardedModel(eng, ego_only=True).forward(inp, pw)
        assert (ego_only is None) == (rank != 0)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_agents_match_single_process(world):
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle.spec import Oracle
    state = export_ptq_state(calibrated_plugin())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, state, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    taps = {}
    want = Oracle(state).forward(scene_np(world), taps)
    single_codes = taps["codes"].reshape(3, world, -1)
    for rank, preds, gathered, calls in results:
        assert gathered.shape == (world, 3, single_codes.shape[-1])
        np.testing.assert_array_equal(gathered, single_codes.transpose(1, 0, 2))          # agent-major wire layout
        shape, a_stride, l_stride, n, ego = calls[0]
        assert (a_stride, l_stride, n, ego) == (3 * single_codes.shape[-1], single_codes.shape[-1], world, rank)
    np.testing.assert_allclose(results[0][1], want["preds_tensor"], rtol=1e-5, atol=1e-6)   # rank 0 = the reference's ego
    # other ranks see the scene from their own pose: a different, finite prediction map of the same shape
    for rank, preds, _, _ in results[1:]:
        assert preds.shape == want["preds_tensor"].shape and np.isfinite(preds).all()
        assert not np.allclose(preds, want["preds_tensor"])


def test_exchange_world_one_is_a_copy():
    from quantv2x_amd.dist import exchange_codes, gathered_strides
    port = _free_port()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        c = torch.arange(3 * 8, dtype=torch.uint8).view(3, 8)
        g = exchange_codes(c)
        assert g.shape == (1, 3, 8) and torch.equal(g[0], c)
        assert gathered_strides(3, 8) == (24, 8)
    finally:
        dist.destroy_process_group()
