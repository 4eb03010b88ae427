"""a7-a11 as ONE launch (qv2x_fuse_heads_batch_f32, round 4): decode + warp + attention + heads tile by tile, the fused map never in HBM.
Same arithmetic as qv2x_fuse_att_batch_f32 followed by qv2x_heads_f32: the predictions and the tapped fused rows are compared BIT FOR BIT
with the two launches (which the oracle-parity suites pin), for one and several agents, ragged scenes, AttFusion and MaxFusion."""
import numpy as np
import pytest
import torch

from _common import calibrated_plugin, scene_np

pytestmark = pytest.mark.gpu


def _batch(scenes):
    """several scenes (different agent counts) as one model call"""
    from quantv2x_amd import synth
    parts, n0 = [], 0
    for sc in scenes:
        part = {k: v.copy() for k, v in sc["inputs_m1"].items()}
        part["voxel_coords"][:, 0] += n0
        parts.append(part)
        n0 += int(sc["record_len"][0])
    L = max(sc["pairwise_t_matrix"].shape[1] for sc in scenes)
    return {"inputs_m1": {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).cuda() for k in parts[0]},
            "agent_modality_list": ["m1"] * n0, "record_len": torch.tensor([int(sc["record_len"][0]) for sc in scenes]),
            "pairwise_t_matrix": torch.from_numpy(np.concatenate([sc["pairwise_t_matrix"] for sc in scenes])).cuda()}


@pytest.mark.parametrize("counts", [[1], [1, 1, 1], [2, 1, 3], [4, 4], [3, 2, 1, 2, 3]])
def test_one_launch_equals_two_launches(counts):
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2))
    eng = deploy(state=state)
    dd = _batch([scene_np(n, seed=11 + i) for i, n in enumerate(counts)])
    eng.fuse_heads_min_tiles = 1 << 30                     # the two launches
    want = {k: v.clone() for k, v in eng(dd).items()}
    taps = {}
    eng(dd, taps)
    eng.fuse_heads_min_tiles, eng.fuse_heads_max_agents = 0, 8     # one launch
    got = eng(dd)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # the tap: the fused rows the one-launch kernel holds in LDS, written out on request, equal the standalone kernel's map
    hw, n_total = eng.fh * eng.fw, sum(counts)
    starts = [sum(counts[:i]) for i in range(len(counts))]
    tap = torch.empty((len(counts), hw, 256), dtype=torch.float32, device="cuda")
    from quantv2x_amd import lib as L
    preds = eng.fuse_heads_scenes(L.ptr(taps["codes"]), hw, n_total * hw, None, dd["pairwise_t_matrix"].to(torch.float64).contiguous(),
                                  [s * hw for s in starts], counts, fused_tap=tap)
    torch.cuda.synchronize()
    assert torch.equal(tap, taps["fused"]) and torch.equal(preds, want["preds_tensor"])


def test_one_launch_max_fusion_and_no_codebook():
    """MaxFusion on a model without the codebook (fp32 shared features instead of code planes): the other branch of fuse_cell_n."""
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2, codebook=False, fusion="max"))
    eng = deploy(state=state)
    dd = _batch([scene_np(3, seed=5), scene_np(2, seed=6)])
    eng.fuse_heads_min_tiles = 1 << 30
    want = {k: v.clone() for k, v in eng(dd).items()}
    eng.fuse_heads_min_tiles, eng.fuse_heads_max_agents = 0, 8
    got = eng(dd)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(got[k], want[k]), k
