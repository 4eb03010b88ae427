"""The UN-QUANTIZED HEAL Pyramid model on the fp32 HIP engine (``quantv2x_amd/engine_pyramid_fp32.py``): the oracle against the torch
mirror / the reference's fp32 vectors (CPU), the engine against the oracle (GPU): every residual block and the wire indices bit-exact
(same fmaf chains), the fused levels and everything after within fp32 tolerance (expf, bilinear taps)."""
import os

import numpy as np
import pytest
import torch

from _common import build_pyramid_plugin, scene, scene_np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pyramid_tiny.npz"))


@pytest.fixture(scope="module")
def fp32_state():
    from quantv2x_amd.engine_pyramid_fp32 import export_fp32_pyramid_state
    torch.set_num_threads(1)
    return export_fp32_pyramid_state(build_pyramid_plugin())


def test_fp32_oracle_matches_the_reference_vectors(fp32_state):
    """oracle (folded BN, fixed summation order) vs the reference's own fp32 forward_with_encdec: same wire indices, predictions to 1e-3"""
    from oracle.spec_fp32 import OraclePyramidFp32
    orc = OraclePyramidFp32(fp32_state)
    taps = {}
    out = orc.forward(scene_np(2), taps)
    assert (taps["codes"].reshape(3, -1) != G["fp32/codes"]).mean() < 2e-3
    d = np.abs(out["preds_tensor"] - G["fp32/preds_tensor"])
    assert np.quantile(d, 0.99) < 2e-2 and out["preds_tensor"].shape == G["fp32/preds_tensor"].shape
    for i in range(3):
        assert out["occ_single_list"][i].shape == G[f"fp32/occ{i}"].shape


@pytest.mark.gpu
@pytest.mark.parametrize("codebook", [True, False])
def test_fp32_engine_matches_the_oracle(codebook):
    from oracle.spec_fp32 import OraclePyramidFp32
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.engine_pyramid_fp32 import DeployedPyramidFp32Model
    model = build_pyramid_plugin(codebook=codebook)
    eng = deploy(model)                                             # a plain model: the fp32 Pyramid engine
    assert isinstance(eng, DeployedPyramidFp32Model)
    orc = OraclePyramidFp32(eng.state)
    for n in (1, 2):
        sc = scene_np(n)
        otaps, gtaps = {}, {}
        want = orc.forward(sc, otaps)
        got = eng(synth.scene_to_torch(sc, "cuda"), gtaps)
        torch.cuda.synchronize()
        checked = 0
        for name, arr in otaps.items():
            if ".resnet.layer" in name:
                np.testing.assert_array_equal(gtaps[name][:, 1:-1, 1:-1].cpu().numpy(), arr, err_msg=name)
                checked += 1
        assert checked == 19
        if codebook:
            np.testing.assert_array_equal(gtaps["codes"].cpu().numpy().reshape(otaps["codes"].shape), otaps["codes"])
        for lvl in range(3):
            np.testing.assert_allclose(gtaps[f"score{lvl}"].cpu().numpy().reshape(otaps[f"score{lvl}"].shape), otaps[f"score{lvl}"], rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(gtaps[f"fused{lvl}"][:, 1:-1, 1:-1].cpu().numpy(), otaps[f"fused{lvl}"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(got["preds_tensor"].cpu().numpy(), want["preds_tensor"], rtol=2e-4, atol=2e-4)
    # and against the torch mirror itself (BN folding + summation order only)
    with torch.no_grad():
        ref = model(scene(2))["preds_tensor"].numpy()
    d = np.abs(got["preds_tensor"].cpu().numpy() - ref)
    assert np.quantile(d, 0.99) < 2e-2
