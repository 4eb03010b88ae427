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
    eng.single_agent_tables = False                        # (single-agent scenes would take the table look-up: test_single_agent_scenes_by_tables)
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


@pytest.mark.parametrize("shape,frames,n_points", [("tiny", 1, 3000), ("tiny", 5, 3000), ("small", 3, 8000)])
def test_single_agent_scenes_by_tables(shape, frames, n_points):
    """qv2x_table_heads_f32: scenes of ONE agent -- AttFusion over one agent is the identity, so every head (cls | reg | dir and the *_single
    ones) is three table rows per cell.  Against the general path (decode + warp + attention, then the GEMM heads) and against the oracle:
    a different fp32 association before the heads' output quantizers, so equal up to rare +-1 LSB flips (the rule every head is held to)."""
    from _common import head_lsb
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin(shape, n_agents=2, n_points=n_points))
    eng = deploy(state=state)
    assert eng.table_heads is not None
    scenes = [scene_np(1, shape=shape, seed=21 + i, n_points=n_points) for i in range(frames)]
    dd = _batch(scenes)
    eng.single_agent_tables, eng.fuse_heads_min_tiles = False, 1 << 30
    want = {k: v.clone() for k, v in eng(dd).items()}
    eng.single_agent_tables = True
    got = eng(dd)
    torch.cuda.synchronize()
    assert set(got) == set(want)
    orc = Oracle(state)
    for k in want:
        lsb = head_lsb(state, "_single" if k.endswith("_single") else "")
        d = (got[k] - want[k]).abs()
        assert float(d.max()) <= lsb * 1.001 and float((d > 1e-5).float().mean()) < 1e-3, (k, float(d.max()))
    for f, sc in enumerate(scenes):
        o = orc.forward(sc)
        for k in ("preds_tensor", "cls_preds_single", "reg_preds_single"):
            lsb = head_lsb(state, "_single" if k.endswith("_single") else "")
            d = np.abs(got[k][f:f + 1].cpu().numpy() - o[k])
            assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 1e-3, (f, k, d.max())
    # a scene of two agents in the batch: the general path takes the whole call
    mixed = _batch([scene_np(1, shape=shape, seed=5, n_points=n_points), scene_np(2, shape=shape, seed=6, n_points=n_points)])
    a = {k: v.clone() for k, v in eng(mixed).items()}
    eng.single_agent_tables = False
    b = eng(mixed)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_single_agent_shortcut_honours_a_non_identity_self_transform():
    """the table look-up skips the warp, so it is only taken when pairwise[b, 0, 0] = I.  A caller that hands a shifted / rotated
    self-transform (pose-noise experiments) gets what the reference computes -- AttFusion warps with t_matrix[0] whatever it is
    (fusion_in_one.py:142-143): the general path, equal to the oracle, for a device tensor and for a CPU tensor."""
    from _common import head_lsb
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2))
    eng = deploy(state=state)
    assert eng.table_heads is not None and eng.single_agent_tables
    sc = scene_np(1, seed=31)
    sc["pairwise_t_matrix"] = sc["pairwise_t_matrix"].copy()
    sc["pairwise_t_matrix"][0, 0, 0] = synth.pose_matrix(1.3, -0.7, 0.2)          # NOT the identity
    want = Oracle(state).forward(sc)
    ident = scene_np(1, seed=31)
    plain = Oracle(state).forward(ident)
    assert np.abs(want["preds_tensor"] - plain["preds_tensor"]).max() > 1e-3       # the warp matters for this scene
    for on_cpu in (False, True):
        dd = _batch([sc])
        if on_cpu:
            dd["pairwise_t_matrix"] = dd["pairwise_t_matrix"].cpu()
        got = eng(dd)
        torch.cuda.synchronize()
        lsb = head_lsb(state, "")
        d = np.abs(got["preds_tensor"].cpu().numpy() - want["preds_tensor"])
        assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 1e-3, (on_cpu, d.max())
    # and the identity still takes the shortcut
    assert eng._self_transforms_are_identity(_batch([ident])["pairwise_t_matrix"])


@pytest.mark.parametrize("c0,c1,kc,hw,n", [(72, 20, 128, 35200, 32), (70, 21, 96, 35200, 32), (0, 8, 32, 35200, 32), (18, 6, 128, 65536, 16), (18, 6, 128, 65536, 17)])
def test_table_heads_at_batch_size_four_cells_per_lane(c0, c1, kc, hw, n):
    """qv2x_table_heads_f32 at the bench's batch (32 V2X-Real frames: the launch takes table_heads4_kernel -- four cells per lane, 16-byte NCHW
    stores, the output quantizer as the division-exact sandwich) against the statement of the same fp32 operations in torch: bias + the
    planes' rows in plane order, then (clamp(rint(y / d) + z, 0, 255) - z) * d on the channels that carry a quantizer.  Bit for bit."""
    import ctypes as C
    from quantv2x_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(7)
    planes = 3                                                        # (65536 x 16: exactly the smallest launch that takes the four-cell form)
    R, CT = n * hw, c0 + c1
    codes = torch.randint(0, kc, (planes, R), dtype=torch.uint8, device="cuda", generator=g)
    tab = torch.randn((planes, kc, CT), device="cuda", generator=g)
    b = torch.randn(CT, device="cuda", generator=g)
    da = torch.rand(CT, device="cuda", generator=g) * 0.05 + 0.01
    da[::5] = -1.0                                                     # channels without an output quantizer
    za = torch.randint(0, 256, (CT,), device="cuda", generator=g).float()
    o0 = torch.empty((n, c0, hw), device="cuda") if c0 else None
    o1 = torch.empty((n, c1, hw), device="cuda")
    L.check(lib.qv2x_table_heads_f32(L.ptr(codes), R, hw, planes, kc, c0, c1, L.ptr(tab), L.ptr(b), L.ptr(da), L.ptr(za),
                                     L.ptr(o0) if c0 else None, L.ptr(o1), L.current_stream()), "table heads")
    torch.cuda.synchronize()
    for lo in range(0, CT, 8):                                         # eight channels at a time: 144 MB per temporary
        hi = min(CT, lo + 8)
        y = b[lo:hi].expand(R, hi - lo).clone()
        for p in range(planes):
            y += tab[p, :, lo:hi][codes[p].long()]
        d, z = da[lo:hi], za[lo:hi]
        q = (torch.clamp(torch.round(y / d) + z, 0, 255) - z) * d
        want = torch.where(d > 0, q, y).reshape(n, hw, hi - lo).permute(0, 2, 1)
        for c in range(lo, hi):
            got = o0[:, c] if c < c0 else o1[:, c - c0]
            assert torch.equal(got, want[:, c - lo]), c
