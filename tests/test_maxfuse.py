"""F-Cooper's ``MaxFusion`` (``fusion_method: max``; hypes_yaml/v2x_real/{LiDAROnly/lidar_fcooper, Codebook/Fcooper}): the plugin mirror and
the oracle against vectors from the reference (``tests/golden/maxfuse.npz``, ``make_golden.py maxfuse``), the HIP engine against the oracle."""
import os

import numpy as np
import pytest
import torch

from _common import build_plugin, calibrated_plugin, compare_frame, hard_forward, scene, scene_np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "maxfuse.npz"))
GEO = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry.npz"))


def test_oracle_and_mirror_match_the_reference():
    from oracle import geometry
    from quantv2x_amd.plugin.models.fuse_modules.fusion_in_one import MaxFusion
    aff = geometry.normalize_pairwise_tfm(GEO["pairwise"], 12.8, 25.6, 1)
    src = np.ascontiguousarray(GEO["src"].transpose(0, 2, 3, 1))
    np.testing.assert_allclose(geometry.max_fuse(geometry.warp_to_ego(src, aff[0], 4)).transpose(2, 0, 1), G["max_fused"][0], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(geometry.max_fuse(geometry.warp_to_ego(src[:1], aff[0], 1)).transpose(2, 0, 1), G["max_fused_n1"][0], rtol=1e-5, atol=2e-6)
    with torch.no_grad():
        got = MaxFusion()(torch.from_numpy(GEO["src"]), torch.tensor([1, 3]), torch.from_numpy(np.concatenate([GEO["affine_f64"]] * 2)))
    np.testing.assert_allclose(got.numpy(), G["max_fused_b2"], rtol=1e-6, atol=1e-6)


def test_model_mirror_matches_the_reference():
    torch.set_num_threads(1)
    model = build_plugin(fusion="max")
    assert list(model.state_dict().keys()) == [str(k) for k in G["state_dict_keys"]]
    with torch.no_grad():
        for n in (1, 2, 3):
            np.testing.assert_allclose(hard_forward(model, scene(n)).numpy(), G[f"preds_tensor_n{n}"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("codebook", [True, False])
def test_hip_engine_max_fusion_vs_oracle(codebook):
    """the whole frame (uint8 activations, indices bit-exact; fused map 2e-5; head LSB) with the fusion kernel in max mode"""
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin(fusion="max", codebook=codebook))
    assert str(state["meta/fusion_method"]) == "max"
    eng, orc = deploy(state=state), Oracle(state)
    for n in (1, 3):
        compare_frame(orc, eng, scene_np(n), state)


@pytest.mark.gpu
def test_hip_fp32_engine_max_fusion():
    from oracle.spec_fp32 import OracleFp32
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.engine_fp32 import export_fp32_state
    st = export_fp32_state(build_plugin(fusion="max"))
    eng = deploy(state=st)
    sc = scene_np(2)
    want = OracleFp32(st).forward(sc)
    got = eng(synth.scene_to_torch(sc, "cuda"))
    np.testing.assert_allclose(got["preds_tensor"].cpu().numpy(), want["preds_tensor"], rtol=2e-4, atol=2e-4)


def test_compressor_model_mirror_matches_the_reference():
    """``NaiveCompressor`` baseline (hypes_yaml/v2x_real/Naive_Compressor/Attfuse: 256 -> 16 -> 256 -> 256 around the link)"""
    torch.set_num_threads(1)
    model = build_plugin(codebook=False, compress_ratio=16)
    assert list(model.state_dict().keys()) == [str(k) for k in G["compress/state_dict_keys"]]
    with torch.no_grad():
        np.testing.assert_allclose(model(scene(2))["preds_tensor"].numpy(), G["compress/preds_tensor_n2"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("fusion", ["att", "max"])
def test_hip_engine_compressor_vs_oracle(fusion):
    """the compressor's three convolutions on the int8 kernels (the 16-channel bottleneck padded to 64): every code bit-exact"""
    from _common import interior_u8
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin(fusion=fusion, codebook=False, compress_ratio=16))
    assert bool(state["meta/compress"])
    eng, orc = deploy(state=state), Oracle(state)
    for n in (1, 2):
        otaps, gtaps, _, _ = compare_frame(orc, eng, scene_np(n), state)
        for name in ("compressor.encoder.0", "compressor.decoder.0", "compressor.decoder.1"):
            want = otaps[name]
            np.testing.assert_array_equal(interior_u8(gtaps[name])[..., :want.shape[-1]], want, err_msg=name)
    assert eng.comp_channels == 16
