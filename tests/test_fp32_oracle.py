"""The fp32 (un-quantized) oracle against the torch mirror of the reference's plain fp32 forward (tiny shape, CPU).

The mirror is pinned on the reference (tests/golden/tiny_fp32.npz, test_plugin_golden.py).  The oracle folds BatchNorm and sums every
dot product in the f32-MFMA kernel's order, so it agrees with the mirror to fp32 re-association: the shared feature within 1e-4
relative, codebook indices on all but near-tie rows, predictions where the codes agree."""
import numpy as np
import torch

from _common import build_plugin, hard_forward, scene, scene_np


def test_fp32_oracle_matches_the_torch_mirror():
    from oracle.spec_fp32 import OracleFp32
    from quantv2x_amd.engine_fp32 import export_fp32_state, pack_k8
    torch.set_num_threads(4)
    model = build_plugin("tiny")
    state = export_fp32_state(model)
    assert str(state["meta/mode"]) == "fp32" and state["backbone_m1.blocks.2.9/w"].shape == (256, 256, 3, 3)
    taps_t, taps_o = {}, {}
    with torch.no_grad():
        want = hard_forward(model, scene(2), taps_t).numpy()
    got = OracleFp32(state).forward(scene_np(2), taps_o)
    sf = taps_t["spatial_features"].numpy().transpose(0, 2, 3, 1)
    np.testing.assert_allclose(taps_o["canvas"], sf, rtol=1e-5, atol=1e-5)
    shr = taps_t["shrinker"].numpy().transpose(0, 2, 3, 1)
    scale = np.abs(shr).max()
    np.testing.assert_allclose(taps_o["shrinker_m1.layers.0.double_conv.1"], shr, rtol=0, atol=2e-4 * scale)
    codes_t = taps_t["codes"].numpy()
    same = (taps_o["codes"] == codes_t)
    assert same.mean() > 0.995, same.mean()
    cells = same.all(axis=0)                                          # cells whose three indices agree ...
    ego = cells[0]                                                    # ... (the single-agent heads see one agent's cells)
    d = np.abs(got["cls_preds_single"][0][:, ego] - model.cls_head_single(taps_t["decoded"]).detach().numpy()[0][:, ego])
    assert d.max() < 1e-4
    assert got["preds_tensor"].shape == want.shape and np.isfinite(got["preds_tensor"]).all()
    # the weight packing: element [g][col][half][e] = W[col][8 g + 4 half + e]
    w = np.arange(64 * 16, dtype=np.float32).reshape(64, 16)
    p = pack_k8(w)
    assert p.shape == (2, 64, 2, 4) and p[1, 5, 1, 2] == w[5, 8 + 4 + 2]
