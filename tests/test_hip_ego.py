"""a7-a11 with ``ego != 0``: what every rank r > 0 of the N-GPU path runs by default (``dist.py``: "every rank is the ego of its own view",
``ego = rank``).  The reference has no ego argument -- its ego is agent 0 of the scene, whose row of the pairwise matrix warps the agents
and whose attention row is kept (opencood/models/fuse_modules/fusion_in_one.py:131-151, ``i = 0 # ego``; ``T[i, j]`` from
opencood/utils/transformation_utils.py:21-66) -- so the reference result for "agent r is the ego" is the SAME scene presented ego-first:
agents ``[r] + others``, the pairwise matrix permuted the same way.  The kernels keep the agents in wire (rank) order and switch the query
and the pairwise row (``csrc/fuse_att.h``): a re-association of the attention's fp32 sums over the agents, inside FUSE_TOL.

Every entry point the N-GPU driver and the tests reach with an ego: ``fuse`` / ``fuse_and_heads`` / ``fuse_heads_and_single`` /
``fuse_frames_and_heads`` (both of its forms: one launch, two launches; the pairwise matrices built on the device from the gathered poses)
and the Pyramid engine's ``decode_features`` -- for EVERY r of a 3- and an 8-agent tiny scene (the 4-agent V2X-Real scene:
``test_hip_fullsize.py::test_v2xreal_every_agent_as_ego``)."""
import numpy as np
import pytest
import torch

from _common import FUSE_TOL, calibrated_plugin, calibrated_pyramid_plugin, ego_first, ego_view, head_lsb, scene_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2))
    return state, deploy(state=state), Oracle(state)


def oracle_view(orc, codes, pairwise_b, n, ego):
    """The oracle on the scene presented ego-first, from the agents' code planes u8 [levels, n, hw]: (fused [h*w, 256], preds [1, C, h, w],
    the ego's own ``*_single`` predictions or None)."""
    lv, _, hw = codes.shape
    h, w = orc.ny // 2, orc.nx // 2
    feats = orc.decode(np.ascontiguousarray(codes).reshape(lv, -1)).reshape(n, h, w, 256)
    f2, t2 = ego_view(feats, pairwise_b, n, ego)
    fused = orc.fuse(f2, t2, [n])
    preds = np.concatenate(orc.heads(fused), axis=1)
    single = np.concatenate(orc.heads(feats[ego:ego + 1], "_single"), axis=1) if bool(orc.s["meta/supervise_single"]) else None
    return fused.reshape(h * w, 256), preds, single


def assert_heads(got, want, lsb, what, flips_floor=2):
    d = np.abs(got - want)
    assert d.max() <= lsb * 1.001 and ((d > 1e-5).mean() < 1e-3 or int((d > 1e-5).sum()) <= flips_floor), (what, d.max(), (d > 1e-5).mean())


def every_entry_point_as_ego(state, eng, orc, sc, n, egos, layout="line"):
    """all the a7-a11 entry points with ego = r against the ego-first oracle; returns the largest fused-map difference seen"""
    from quantv2x_amd import lib as L, synth
    from quantv2x_amd.dist import payload_layout, POSE_BYTES
    taps = {}
    eng(synth.scene_to_torch(sc, "cuda"), taps)
    hw = eng.fh * eng.fw
    codes = taps["codes"].clone().view(eng.levels, n, hw)
    codes_np = codes.cpu().numpy()
    pairwise_b = torch.from_numpy(sc["pairwise_t_matrix"][0]).cuda().contiguous()
    lsb, lsb_s = head_lsb(state), head_lsb(state, "_single")
    worst = 0.0
    for r in egos:
        want_fused, want_preds, want_single = oracle_view(orc, codes_np, sc["pairwise_t_matrix"][0], n, r)
        fused = torch.empty((hw, 256), dtype=torch.float32, device="cuda")
        eng.fuse(L.ptr(codes), hw, n * hw, None, pairwise_b, n, fused, ego=r)
        torch.cuda.synchronize()
        np.testing.assert_allclose(fused.cpu().numpy(), want_fused, err_msg=f"fused map, ego {r}", **FUSE_TOL)
        worst = max(worst, float(np.abs(fused.cpu().numpy() - want_fused).max()))
        out = eng.fuse_and_heads(codes, hw, n * hw, pairwise_b, n, ego=r)
        assert_heads(out["preds_tensor"].cpu().numpy(), want_preds, lsb, ("fuse_and_heads", r))
        own = codes[:, r:r + 1].contiguous()
        out = eng.fuse_heads_and_single(codes, hw, n * hw, pairwise_b, n, r, own)
        assert_heads(out["preds_tensor"].cpu().numpy(), want_preds, lsb, ("fuse_heads_and_single", r))
        sp = torch.cat([out["cls_preds_single"], out["reg_preds_single"], out["dir_preds_single"]], dim=1).cpu().numpy()
        assert_heads(sp, want_single, lsb_s, ("single heads of the ego's own codes", r))
    # the N-GPU driver's call: ``frames`` scenes in the gathered wire layout (agent-major payloads: code planes [levels, frames, hw], then
    # the poses), the pairwise matrices built on the device from the gathered poses.  Frame 1 = the same agents in reversed wire order.
    frames = 2
    cbytes, pose_off, pbytes = payload_layout(eng.levels, frames, hw)
    poses = synth.agent_poses(n, layout)                            # the scene's own poses (synth.make_scene)
    gathered = torch.zeros((n, pbytes), dtype=torch.uint8, device="cuda")
    per_frame = [codes_np, codes_np[:, ::-1].copy()]
    for a in range(n):
        planes = np.stack([per_frame[f][:, a] for f in range(frames)], axis=1)               # [levels, frames, hw]
        gathered[a, :cbytes] = torch.from_numpy(planes.reshape(-1)).cuda()
        gathered[a, pose_off:] = torch.from_numpy(np.tile(poses[a].reshape(-1), frames).view(np.uint8)).cuda()
    L_ = sc["pairwise_t_matrix"].shape[1]
    pairwise = torch.zeros((frames, L_, L_, 4, 4), dtype=torch.float64, device="cuda")
    eng.pairwise_frames_from_poses(gathered, n, pbytes, pose_off, frames, POSE_BYTES, L_, pairwise)
    torch.cuda.synchronize()
    pw = pairwise.cpu().numpy()
    np.testing.assert_allclose(pw[0], sc["pairwise_t_matrix"][0], rtol=0, atol=1e-12)
    saved = eng.fuse_heads_min_tiles, eng.fuse_heads_max_agents
    try:
        for form, (tiles, agents) in {"one launch": (0, 8), "two launches": (1 << 30, saved[1])}.items():
            eng.fuse_heads_min_tiles, eng.fuse_heads_max_agents = tiles, agents
            for r in egos:
                own = torch.from_numpy(np.stack([per_frame[f][:, r] for f in range(frames)], axis=1)).cuda().contiguous()
                out = eng.fuse_frames_and_heads(gathered, pbytes, frames * hw, hw, pairwise, n, r, own, frames)
                torch.cuda.synchronize()
                for f in range(frames):
                    _, want_preds, want_single = oracle_view(orc, per_frame[f], pw[f], n, r)
                    assert_heads(out["preds_tensor"][f:f + 1].cpu().numpy(), want_preds, lsb, (form, "frame", f, "ego", r))
                    sp = torch.cat([out["cls_preds_single"], out["reg_preds_single"], out["dir_preds_single"]], dim=1)[f:f + 1].cpu().numpy()
                    assert_heads(sp, want_single, lsb_s, (form, "single", f, r))
    finally:
        eng.fuse_heads_min_tiles, eng.fuse_heads_max_agents = saved
    return worst


@pytest.mark.parametrize("n_agents", [3, 8])
def test_tiny_scene_every_agent_as_ego(tiny, n_agents):
    state, eng, orc = tiny
    sc = scene_np(n_agents, seed=9)
    every_entry_point_as_ego(state, eng, orc, sc, n_agents, range(n_agents))


def test_the_ego_changes_the_result_and_ego_zero_is_the_plain_call(tiny):
    """guards the test itself: the views differ from one another by far more than the tolerance; ego = 0 is the default call"""
    from quantv2x_amd import lib as L, synth
    state, eng, orc = tiny
    n, sc = 3, scene_np(3, seed=9)
    taps = {}
    plain = eng(synth.scene_to_torch(sc, "cuda"), taps)["preds_tensor"].clone()
    hw = eng.fh * eng.fw
    pairwise_b = torch.from_numpy(sc["pairwise_t_matrix"][0]).cuda().contiguous()
    maps = []
    for r in range(n):
        fused = torch.empty((hw, 256), dtype=torch.float32, device="cuda")
        eng.fuse(L.ptr(taps["codes"]), hw, n * hw, None, pairwise_b, n, fused, ego=r)
        maps.append(fused.cpu().numpy())
    assert np.array_equal(maps[0], taps["fused"].cpu().numpy().reshape(hw, 256))
    assert min(np.abs(maps[0] - maps[1]).max(), np.abs(maps[1] - maps[2]).max()) > 1e-2
    assert torch.equal(eng.fuse_and_heads(taps["codes"], hw, n * hw, pairwise_b, n, ego=0)["preds_tensor"], plain)


def test_max_fusion_with_an_ego(tiny):
    """MaxFusion (fusion_in_one.py:100-124 warps with row 0 as well): order-independent, so the ego-first oracle must match to the warp's tolerance"""
    from oracle.spec import Oracle
    from quantv2x_amd import lib as L, synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2, fusion="max"))
    eng, orc = deploy(state=state), Oracle(state)
    n, sc = 3, scene_np(3, seed=4)
    taps = {}
    eng(synth.scene_to_torch(sc, "cuda"), taps)
    hw = eng.fh * eng.fw
    codes = taps["codes"].clone().view(eng.levels, n, hw)
    pairwise_b = torch.from_numpy(sc["pairwise_t_matrix"][0]).cuda().contiguous()
    for r in range(n):
        want_fused, want_preds, _ = oracle_view(orc, codes.cpu().numpy(), sc["pairwise_t_matrix"][0], n, r)
        fused = torch.empty((hw, 256), dtype=torch.float32, device="cuda")
        eng.fuse(L.ptr(codes), hw, n * hw, None, pairwise_b, n, fused, ego=r)
        np.testing.assert_allclose(fused.cpu().numpy(), want_fused, **FUSE_TOL)
        assert_heads(eng.fuse_and_heads(codes, hw, n * hw, pairwise_b, n, ego=r)["preds_tensor"].cpu().numpy(), want_preds, head_lsb(state), r)


# ---- the Pyramid engine (pyramid_fuse.py:17-62: weighted_fuse warps with t_matrix[0] and sums in agent order) -----------------------------
@pytest.mark.parametrize("n_agents", [2, 3])
def test_pyramid_decode_features_every_agent_as_ego(n_agents):
    from oracle.spec_pyramid import OraclePyramid
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    from test_hip_pyramid import check_after_fuse
    st = export_ptq_state(calibrated_pyramid_plugin())
    orc, eng = OraclePyramid(st), deploy(state=st)
    sc = scene_np(n_agents, seed=6)
    dd = synth.scene_to_torch(sc, "cuda")
    codes = eng.encode_features(dd["inputs_m1"], n_agents).clone()                    # [levels, n, hw]
    hw = eng.fh * eng.fw
    pairwise = dd["pairwise_t_matrix"].to(torch.float64).contiguous()
    feats = orc.decode(codes.cpu().numpy().reshape(codes.shape[0], -1)).reshape(n_agents, eng.fh, eng.fw, orc.D)
    seen = []
    for r in range(n_agents):
        gtaps, otaps = {}, {}
        got = eng.decode_features(codes, hw, n_agents * hw, [n_agents], pairwise, ego=r, taps=gtaps)
        torch.cuda.synchronize()
        f2, t2 = ego_view(feats, sc["pairwise_t_matrix"][0], n_agents, r)
        _, _, occs = orc.pyramid(f2, t2, [n_agents], otaps)
        order = ego_first(n_agents, r)
        for lvl in range(3):
            fu = otaps[f"fused{lvl}"]
            np.testing.assert_allclose(gtaps[f"fused{lvl}"].cpu().numpy().reshape(fu.shape), fu, err_msg=f"level {lvl}, ego {r}", **FUSE_TOL)
            # the occupancy maps are per agent (no ego in them): the engine's, in wire order, are the oracle's ego-first ones un-permuted
            np.testing.assert_array_equal(got["occ_single_list"][lvl].cpu().numpy()[order], occs[lvl])
        check_after_fuse(orc, st, gtaps, got, otaps)                                          # deblocks, shrink_conv, heads on the engine's fused maps
        seen.append(gtaps["fused0"].cpu().numpy())
    assert np.abs(seen[0] - seen[1]).max() > 1e-2
