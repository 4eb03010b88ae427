"""f3: the HEAL Pyramid-fusion path on the HIP engine (C ABI) against the CPU oracle (``oracle/spec_pyramid.py``), stage by stage.

Everything up to the occupancy scores is integer / fixed-order fp32 arithmetic and must agree BIT FOR BIT over the whole frame
(canvas, every residual block of the agent and of the three ResNeXt levels, the wire indices, the decoded map, occupancy codes and
scores).  ``weighted_fuse`` (bilinear taps, expf) agrees within FUSE_TOL; everything after it is checked on the engine's own fused
maps (oracle stage fed the GPU's input), where it is again bit-exact / within one LSB of the head quantizer."""
import numpy as np
import pytest
import torch

from _common import FUSE_TOL, calibrated_pyramid_plugin, interior_u8, scene_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle.spec_pyramid import OraclePyramid
    st = export_ptq_state(calibrated_pyramid_plugin())
    return st, OraclePyramid(st), deploy(state=st)


def check_after_fuse(orc, st, gtaps, got, otaps):
    """Everything after ``weighted_fuse`` on the ENGINE's own fused maps (the oracle stage fed the GPU's input): the deblocks and
    shrink_conv bit-exact, the heads within one LSB of their quantizer; the deblock codes also against the ones of the oracle's own
    fused maps in ``otaps`` (rare +-1 flips)."""
    cat_q, c0 = [], 0
    for lvl in range(3):
        name = f"pyramid_backbone.deblocks.{lvl}.0"
        gf = gtaps[f"fused{lvl}"].cpu().numpy().reshape(otaps[f"fused{lvl}"].shape)
        ref, oq = orc.dense_f32in(name, gf, orc.ups[lvl])
        np.testing.assert_array_equal(interior_u8(gtaps["cat"])[..., c0:c0 + 128], ref, err_msg=name)
        cat_q.append((c0, 128, oq[0], oq[1]))
        c0 += 128
        flips = (ref != otaps[name]).mean()
        assert flips < 5e-3, (name, flips)                              # vs the oracle's own fused map: rare +-1 flips
    gcat = interior_u8(gtaps["cat"])
    taps2 = {}
    shr, shr_q = orc.shrink(gcat, cat_q, taps2)
    for nme in ("shrink_conv.layers.0.double_conv.0", "shrink_conv.layers.0.double_conv.1"):
        np.testing.assert_array_equal(interior_u8(gtaps[nme]), taps2[nme], err_msg=nme)
    cls, reg, dr = orc.heads(orc.dequant(shr, shr_q))
    ref_preds = np.concatenate([cls, reg, dr], axis=1)
    lsb = max(float(st[h + "/a_delta"]) for h in ("cls_head", "reg_head", "dir_head"))
    d = np.abs(got["preds_tensor"].cpu().numpy() - ref_preds)
    assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 1e-3, (d.max(), (d > 1e-5).mean())
    for k, sl in (("cls_preds", slice(0, 18)), ("reg_preds", slice(18, 60)), ("dir_preds", slice(60, 72))):
        assert torch.equal(got[k], got["preds_tensor"][:, sl])


def compare_pyramid_frame(orc, eng, sc, st):
    from quantv2x_amd import synth
    from oracle import geometry
    otaps, gtaps = {}, {}
    want = orc.forward(sc, otaps)
    got = eng(synth.scene_to_torch(sc, "cuda"), gtaps)
    torch.cuda.synchronize()
    n = len(sc["agent_modality_list"])
    np.testing.assert_array_equal(interior_u8(gtaps["canvas"]), otaps["canvas"], err_msg="canvas")
    checked = 0
    for name, arr in otaps.items():
        if ".resnet.layer" in name and name in gtaps:
            np.testing.assert_array_equal(interior_u8(gtaps[name]), arr, err_msg=name)
            checked += 1
    assert checked >= 3 * 2 + 16
    if orc.has_codebook:
        np.testing.assert_array_equal(gtaps["codes"].cpu().numpy().reshape(otaps["codes"].shape), otaps["codes"], err_msg="wire indices")
        np.testing.assert_array_equal(gtaps["features"].cpu().numpy().reshape(otaps["features"].shape), otaps["features"], err_msg="decoded map")
    H, W = (float(v) for v in st["meta/HW_metres"])
    affine = geometry.normalize_pairwise_tfm(np.asarray(sc["pairwise_t_matrix"]), H, W, float(st["meta/discrete_ratio"]))
    lens = [int(v) for v in sc["record_len"]]
    for lvl in range(3):
        oc, sco, fu = otaps[f"occ_code{lvl}"], otaps[f"score{lvl}"], otaps[f"fused{lvl}"]
        np.testing.assert_array_equal(gtaps[f"occ_code{lvl}"].cpu().numpy().reshape(oc.shape), oc, err_msg=f"occupancy codes {lvl}")
        np.testing.assert_array_equal(gtaps[f"score{lvl}"].cpu().numpy().reshape(sco.shape), sco, err_msg=f"score {lvl}")
        gf = gtaps[f"fused{lvl}"].cpu().numpy().reshape(fu.shape)
        np.testing.assert_allclose(gf, fu, **FUSE_TOL)
        np.testing.assert_allclose(got["occ_single_list"][lvl].cpu().numpy(), want["occ_single_list"][lvl], rtol=0, atol=0)
    check_after_fuse(orc, st, gtaps, got, otaps)
    return want, got


@pytest.mark.parametrize("n_agents", [1, 2, 3])
def test_tiny_frame_every_stage(tiny, n_agents):
    st, orc, eng = tiny
    compare_pyramid_frame(orc, eng, scene_np(n_agents), st)


def test_small_frame_every_stage():
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle.spec_pyramid import OraclePyramid
    st = export_ptq_state(calibrated_pyramid_plugin("small", n_points=8000))
    compare_pyramid_frame(OraclePyramid(st), deploy(state=st), scene_np(2, "small", n_points=8000), st)


def test_model_without_codebook():
    """``heter_pyramid_collab_mc`` (hypes_yaml/v2x_real/LiDAROnly/lidar_pyramid.yaml): the agents' activation codes go straight into
    the pyramid -- every stage against the oracle; there is no wire format, so the multi-GPU driver and the encdec entry points refuse."""
    from quantv2x_amd import lib as L
    from quantv2x_amd.dist import AgentShardedModel
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle.spec_pyramid import OraclePyramid
    st = export_ptq_state(calibrated_pyramid_plugin(codebook=False))
    assert not bool(st["meta/has_codebook"])
    eng = deploy(state=st)
    for n in (1, 2):
        compare_pyramid_frame(OraclePyramid(st), eng, scene_np(n), st)
    with pytest.raises(L.Qv2xError):
        eng.encode_features({}, 1)
    with pytest.raises(NotImplementedError):
        AgentShardedModel(eng)


def test_encdec_split_and_reference_model_contract(tiny):
    """encode_features -> (the wire) -> decode_features equals forward; the plugin's fake-quant model agrees on the detections' level."""
    from quantv2x_amd import synth
    st, orc, eng = tiny
    sc = scene_np(2)
    dd = synth.scene_to_torch(sc, "cuda")
    whole = eng(dd)["preds_tensor"].clone()
    codes = eng.encode_features(dd["inputs_m1"], 2).clone()
    hw = eng.fh * eng.fw
    assert codes.shape == (3, 2, hw) and codes.dtype == torch.uint8 and int(codes.max()) < eng.kc
    split = eng.decode_features(codes, hw, 2 * hw, [2], dd["pairwise_t_matrix"].to(torch.float64).contiguous())
    assert torch.equal(split["preds_tensor"], whole)
    assert split["pyramid"] == "collab" and len(split["occ_single_list"]) == 3
    replay = eng.capture(dd)
    assert torch.equal(replay()["preds_tensor"], whole)


def test_deploy_dispatch_and_refusals(tiny):
    from quantv2x_amd.engine import DeployedModel, deploy
    from quantv2x_amd.engine_pyramid import DeployedPyramidModel
    st, _, eng = tiny
    assert isinstance(eng, DeployedPyramidModel)
    with pytest.raises(NotImplementedError):
        DeployedModel(st)
    dd = {"agent_modality_list": ["m2"], "pairwise_t_matrix": torch.eye(4).view(1, 1, 1, 4, 4), "inputs_m1": {}}
    with pytest.raises(NotImplementedError):
        eng(dd)


def test_v2xreal_full_size_two_agents():
    """The V2X-Real grid (704 x 200 pillars, 60k points per agent, 2 agents): every stage against the oracle as on the tiny frame --
    3 x 2 x 35 200 wire indices and every activation code of the 19 residual blocks bit-exact -- plus size-independent properties:
    determinism, and the ego's own map is what a lone agent gets when the other agent's occupancy is masked out of range."""
    import copy
    import os
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    from oracle.spec_pyramid import OraclePyramid
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    model = train_utils.create_model(copy.deepcopy(synth.make_pyramid_hypes("v2xreal"))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    calib = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=1, seed=3, n_points=60000))
    st = export_ptq_state(inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib]))
    eng = deploy(state=st)
    sc = synth.make_scene("v2xreal", n_agents=2, seed=3, n_points=60000)
    want, got = compare_pyramid_frame(OraclePyramid(st), eng, sc, st)
    assert got["preds_tensor"].shape == (1, 72, 100, 352)
    dd = synth.scene_to_torch(sc, "cuda")
    a = eng(dd)["preds_tensor"].clone()
    b = eng(dd)["preds_tensor"]
    assert torch.equal(a, b)
    # round trip through the wire format: encode_features once, decode_features twice (cached codes) -> identical detections
    codes = eng.encode_features(dd["inputs_m1"], 2).clone()
    hw = eng.fh * eng.fw
    pw = dd["pairwise_t_matrix"].to(torch.float64).contiguous()
    c = eng.decode_features(codes, hw, 2 * hw, [2], pw)["preds_tensor"]
    assert torch.equal(a, c)


def test_fused_bottleneck_equals_the_per_layer_kernels(tiny):
    """qv2x_bottleneck_i8 (conv1 -> grouped 3x3 -> conv3 + shortcut in one launch) at every width, against the three separate launches:
    identical codes for every block of the frame"""
    from quantv2x_amd import synth
    st, orc, eng = tiny
    dd = synth.scene_to_torch(scene_np(2), "cuda")
    keep = (eng.fuse_blocks, eng.fuse_planes)
    try:
        eng.fuse_blocks, eng.fuse_planes = True, (64, 128, 256)
        a = {}
        pa = eng(dd, a)["preds_tensor"].clone()
        eng.fuse_blocks = False
        b = {}
        pb = eng(dd, b)["preds_tensor"]
        torch.cuda.synchronize()
        names = [k for k in a if ".resnet.layer" in k and k.startswith("pyramid_backbone")]
        assert len(names) == 16
        for k in names:
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(pa, pb)
    finally:
        eng.fuse_blocks, eng.fuse_planes = keep
