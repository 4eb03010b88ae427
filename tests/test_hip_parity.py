"""HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.  Needs an MI355X: ``-m gpu``.

Bar: bit-exact for every uint8 activation code and every codebook index; fp32 tolerance (written below) for the
decode / warp / attention / heads outputs."""
import numpy as np
import pytest
import torch

from _common import FUSE_TOL, calibrated_plugin, compare_frame, scene, scene_np

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


def _interior(t):
    """padded i8 BEV [N, H+2, W+2, C] -> uint8 codes [N, H, W, C]"""
    return (t[:, 1:-1, 1:-1, :].to(torch.int16) + 128).to(torch.uint8).cpu().numpy()


@pytest.fixture(scope="module")
def tiny():
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin())
    return state, Oracle(state), deploy(state=state)


def _compare(orc, eng, sc_np, n_agents, state):
    compare_frame(orc, eng, sc_np, state)


@pytest.mark.parametrize("n_agents", [1, 2, 3])
def test_tiny_end_to_end(tiny, n_agents):
    state, orc, eng = tiny
    _compare(orc, eng, scene_np(n_agents), n_agents, state)


@pytest.mark.parametrize("n_agents", [1, 3])
def test_tiny_layer_by_layer_launches(tiny, n_agents):
    """The shipped plan runs backbone level 0 as one fused launch (conv_i8_chain.hip), whose intermediate maps never reach
    HBM.  With the chains off every layer is its own launch and EVERY layer's uint8 map is compared with the oracle; the
    two plans must agree bit for bit on everything downstream."""
    state, orc, eng = tiny
    sc = scene_np(n_agents)
    assert eng.use_chains and eng.chains[0] is not None
    keep, eng.chain_max_agents = eng.chain_max_agents, 8           # (the engine itself fuses for one agent-frame only)
    try:
        _, fused_taps, _, fused_out = compare_frame(orc, eng, sc, state)
    finally:
        eng.chain_max_agents = keep
    fused_out = {k: v.clone() for k, v in fused_out.items()}
    eng.use_chains = False
    try:
        _, taps, _, out = compare_frame(orc, eng, sc, state)
        assert "backbone_m1.blocks.0.2" in taps and "backbone_m1.blocks.0.2" not in fused_taps
        for k in ("backbone_m1.blocks.0.4", "cat", "codes"):
            assert torch.equal(taps[k], fused_taps[k]), k
        for k in out:
            assert torch.equal(out[k], fused_out[k]), k
    finally:
        eng.use_chains = True


def test_tiny_ragged_pillars(tiny):
    """pillars with 1..32 points, the 32-point cap, and an agent whose sweep is almost empty"""
    state, orc, eng = tiny
    sc = scene_np(2, n_points=20000)          # dense: many pillars hit the 32-point cap
    assert (sc["inputs_m1"]["voxel_num_points"] == 32).any() and (sc["inputs_m1"]["voxel_num_points"] == 1).any()
    _compare(orc, eng, sc, 2, state)
    sparse = scene_np(2, n_points=40)
    _compare(orc, eng, sparse, 2, state)


def test_graph_replay_matches_eager(tiny):
    state, orc, eng = tiny
    dd = scene(2, device="cuda")
    eager = {k: v.clone() for k, v in eng(dd).items()}
    replay = eng.capture(dd)
    out = replay()
    torch.cuda.synchronize()
    for k in eager:
        assert torch.equal(eager[k], out[k]), k


def test_error_paths(tiny):
    from quantv2x_amd import lib as L
    import ctypes as C
    state, orc, eng = tiny
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin_total, d.stride, d.cout, d.ngroups = 1, 8, 8, 64, 3, 64, 1
    rc = eng.lib.qv2x_conv3x3_i8(C.byref(d), None, None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in eng.lib.qv2x_last_error()
    x = torch.zeros(16, dtype=torch.int8, device="cuda")
    rc = eng.lib.qv2x_conv3x3_i8(C.byref(d), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), None)
    assert rc == -1 and b"stride" in eng.lib.qv2x_last_error()
    with pytest.raises(NotImplementedError):
        dd = scene(2, device="cuda")
        dd["agent_modality_list"] = ["m1", "m2"]
        eng(dd)


def test_model_without_codebook_single_class():
    """HeterModelBaseline (no codebook, single-class heads): the shared feature is the dequantized shrinker output."""
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin(multiclass=False, codebook=False))
    assert not bool(state["meta/has_codebook"])
    eng = deploy(state=state)
    sc = scene_np(3)
    from quantv2x_amd import synth
    ot, gt = {}, {}
    want = Oracle(state).forward(sc, ot)
    got = eng(synth.scene_to_torch(sc, "cuda"), gt)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_interior(gt["shrinker_m1.layers.0.double_conv.1"]), ot["shrinker_m1.layers.0.double_conv.1"])
    np.testing.assert_array_equal(gt["features"].cpu().numpy().reshape(ot["features"].shape), ot["features"])   # exact dequantization
    np.testing.assert_allclose(gt["fused"].cpu().numpy().reshape(ot["fused"].shape), ot["fused"], **FUSE_TOL)
    assert got["preds_tensor"].shape == (1, 20, 16, 32)
    lsb = max(float(state[k + "/a_delta"]) for k in ("cls_head", "reg_head", "dir_head"))
    d = np.abs(got["preds_tensor"].cpu().numpy() - want["preds_tensor"])
    assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 2e-3


def test_empty_agent_and_agent_count_limits(tiny):
    """an agent with no pillar at all (canvas = code of 0.0 everywhere), 8 agents (the kernel's maximum), 9 -> error"""
    from quantv2x_amd import lib as L, synth
    state, orc, eng = tiny
    sc = scene_np(2)
    keep = sc["inputs_m1"]["voxel_coords"][:, 0] == 0          # drop every pillar of agent 1
    sc["inputs_m1"] = {k: v[keep] for k, v in sc["inputs_m1"].items()}
    ot, gt = {}, {}
    orc.forward(sc, ot)
    # the oracle sizes its canvas from the agent list, the engine too
    got = eng(synth.scene_to_torch(sc, "cuda"), gt)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_interior(gt["canvas"]), ot["canvas"])
    np.testing.assert_array_equal(gt["codes"].cpu().numpy().reshape(ot["codes"].shape), ot["codes"])
    sc8 = synth.make_scene("tiny", n_agents=8, seed=11, n_points=1500, layout="ring", max_cav=8)
    o8, g8 = {}, {}
    want = orc.forward(sc8, o8)
    got = eng(synth.scene_to_torch(sc8, "cuda"), g8)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(g8["codes"].cpu().numpy().reshape(o8["codes"].shape), o8["codes"])
    np.testing.assert_allclose(g8["fused"].cpu().numpy().reshape(o8["fused"].shape), o8["fused"], **FUSE_TOL)
    sc9 = synth.make_scene("tiny", n_agents=9, seed=11, n_points=500, max_cav=9)
    with pytest.raises(L.Qv2xError):
        eng(synth.scene_to_torch(sc9, "cuda"))


@pytest.mark.parametrize("n_agents", [1, 3])
def test_sharded_stage_functions_equal_forward(tiny, n_agents):
    """``forward`` runs the two head passes as one launch; the multi-GPU driver (quantv2x_amd/dist.py) calls
    ``encode_agents`` / ``fuse_and_heads`` / ``single_preds`` one by one.  Same frame, bit-identical outputs."""
    from quantv2x_amd import synth
    state, orc, eng = tiny
    dd = synth.scene_to_torch(scene_np(n_agents), "cuda")
    want = {k: v.clone() for k, v in eng(dd).items()}
    codes = eng.encode_agents(dd["inputs_m1"], n_agents).clone()           # [levels, n, hw]
    levels, n, hw = codes.shape
    got = eng.fuse_and_heads(codes, hw, n * hw, dd["pairwise_t_matrix"][0].contiguous(), n_agents, 0)
    got.update(eng.single_preds(codes, n_agents))
    both = eng.fuse_heads_and_single(codes, hw, n * hw, dd["pairwise_t_matrix"][0].contiguous(), n_agents, 0, codes, n_agents)
    torch.cuda.synchronize()
    for key in ("preds_tensor", "cls_preds", "reg_preds", "dir_preds", "cls_preds_single", "reg_preds_single", "dir_preds_single"):
        assert torch.equal(got[key], want[key]), key
        assert torch.equal(both[key], want[key]), key


def test_padding_pillar_rows_are_dropped(tiny):
    """A fixed pillar count M (HIP-graph replay, ``DeployedModel.capture``) is kept by padding the pillar arrays with rows whose
    agent index is out of range (-1) and whose point count is 0: ``pfn_scatter_kernel`` drops them, every output is unchanged."""
    from quantv2x_amd import synth
    state, orc, eng = tiny
    sc = scene_np(2)
    dd = synth.scene_to_torch(sc, "cuda")
    want = {k: v.clone() for k, v in eng(dd).items()}
    pad = 37
    inp = dd["inputs_m1"]
    vf = torch.cat([inp["voxel_features"], torch.zeros((pad, 32, 4), device="cuda")])
    co = torch.cat([inp["voxel_coords"], torch.tensor([[-1, 0, 3, 5]] * pad, dtype=inp["voxel_coords"].dtype, device="cuda")])
    npt = torch.cat([inp["voxel_num_points"], torch.zeros(pad, dtype=inp["voxel_num_points"].dtype, device="cuda")])
    perm = torch.randperm(vf.shape[0], generator=torch.Generator().manual_seed(0)).cuda()       # padding anywhere, not only at the end
    dd2 = dict(dd, inputs_m1={"voxel_features": vf[perm].contiguous(), "voxel_coords": co[perm].contiguous(), "voxel_num_points": npt[perm].contiguous()})
    got = eng(dd2)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_scenes_of_different_sizes_in_one_call(tiny):
    """Two scenes with 3 and 1 agents in one ``forward`` (record_len = [3, 1]): the fusion of both runs as ONE launch
    (qv2x_fuse_att_batch_f32: per-scene code offsets and agent counts) and must give what each scene gives alone, bit for bit."""
    from quantv2x_amd import synth
    _, _, eng = tiny
    a, b = scene_np(3), scene_np(1, seed=11)
    alone = []
    for sc in (a, b):
        taps = {}
        out = eng(synth.scene_to_torch(sc, "cuda"), taps)
        torch.cuda.synchronize()
        alone.append(({k: v.clone() for k, v in out.items()}, taps["fused"].clone()))
    inp = {}
    for k in a["inputs_m1"]:
        pb = b["inputs_m1"][k].copy()
        if k == "voxel_coords":
            pb[:, 0] += 3
        inp[k] = torch.from_numpy(np.concatenate([a["inputs_m1"][k], pb])).cuda()
    both = {"inputs_m1": inp, "agent_modality_list": ["m1"] * 4, "record_len": torch.tensor([3, 1], dtype=torch.int64),
            "pairwise_t_matrix": torch.from_numpy(np.concatenate([a["pairwise_t_matrix"], b["pairwise_t_matrix"]])).cuda()}
    taps = {}
    out = eng(both, taps)
    torch.cuda.synchronize()
    for s, (o1, f1) in enumerate(alone):
        assert torch.equal(taps["fused"][s], f1[0]), f"fused map of scene {s}"
        for k in ("cls_preds", "reg_preds", "dir_preds"):
            assert torch.equal(out[k][s], o1[k][0]), (k, s)
    assert torch.equal(out["cls_preds_single"][:3], alone[0][0]["cls_preds_single"]) and torch.equal(out["cls_preds_single"][3:], alone[1][0]["cls_preds_single"])


def test_resident_canvas_is_clean_between_frames(tiny):
    """The pillar canvas is not re-filled per frame: a frame's cells are set back after the first convolution has read them
    (qv2x_pfn_unscatter_i8).  Different sweeps through the same workspace, eagerly and as a replayed HIP graph, must each match the oracle
    from the canvas on."""
    from quantv2x_amd import synth
    state, orc, eng = tiny
    for seed in (21, 22, 23):
        compare_frame(orc, eng, scene_np(2, seed=seed), state)
    a, b = scene_np(2, seed=31), scene_np(2, seed=32)
    da = synth.scene_to_torch(a, "cuda")
    replay = eng.capture(da)
    want = {}
    for sc in (a, b):
        taps = {}
        orc.forward(sc, taps)
        want[id(sc)] = taps["codes"]
    replay(); torch.cuda.synchronize()
    got_a = eng._workspace(2)["codes"].cpu().numpy().reshape(want[id(a)].shape)
    np.testing.assert_array_equal(got_a, want[id(a)])
    db = synth.scene_to_torch(b, "cuda")
    if db["inputs_m1"]["voxel_features"].shape == da["inputs_m1"]["voxel_features"].shape:     # same pillar count: refresh the inputs in place
        for k in da["inputs_m1"]:
            da["inputs_m1"][k].copy_(db["inputs_m1"][k])
        replay(); torch.cuda.synchronize()
        got_b = eng._workspace(2)["codes"].cpu().numpy().reshape(want[id(b)].shape)
        np.testing.assert_array_equal(got_b, want[id(b)])
    replay(); replay(); torch.cuda.synchronize()
