"""The plugin mirror of the HEAL Pyramid-fusion model (``heter_pyramid_collab_codebook_mc_encdec``) and of its ``QuantModel`` twin
against vectors captured from the reference (``tests/golden/pyramid_tiny.npz``, made by ``make_golden.py pyramid_model``)."""
import os

import numpy as np
import pytest
import torch

from _common import build_pyramid_plugin, calibrated_pyramid_plugin, scene

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pyramid_tiny.npz"))
TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def fp_model():
    torch.set_num_threads(1)
    return build_pyramid_plugin()


@pytest.fixture(scope="module")
def qt():
    torch.set_num_threads(1)
    return calibrated_pyramid_plugin()


def test_state_dict_keys_are_the_reference_checkpoint_keys(fp_model):
    assert list(fp_model.state_dict().keys()) == [str(k) for k in G["state_dict_keys"]]


def test_fp32_forward_hard_and_soft(fp_model):
    dd = scene(2)
    with torch.no_grad():
        o = fp_model.forward_with_encdec(dd)
        codes, _, info = fp_model.encode_features(dd)
        torch.manual_seed(0)
        fp_model.hard_eval = False
        soft = fp_model(dd)
        fp_model.hard_eval = True
        hard_via_forward = fp_model(dd)
        n1 = fp_model.forward_with_encdec(scene(1))
    np.testing.assert_allclose(o["preds_tensor"].numpy(), G["fp32/preds_tensor"], **TOL)
    np.testing.assert_allclose(hard_via_forward["preds_tensor"].numpy(), G["fp32/preds_tensor"], **TOL)
    for i, occ in enumerate(o["occ_single_list"]):
        np.testing.assert_allclose(occ.numpy(), G[f"fp32/occ{i}"], **TOL)
    np.testing.assert_array_equal(torch.stack([c[:, 0] for c in codes]).numpy().astype(np.uint8), G["fp32/codes"])
    np.testing.assert_allclose(info["affine_matrix"].numpy(), G["fp32/affine"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(soft["preds_tensor"].numpy(), G["fp32/soft_preds_tensor_seed0"], **TOL)
    np.testing.assert_allclose(n1["preds_tensor"].numpy(), G["fp32/preds_tensor_n1"], **TOL)
    assert o["pyramid"] == "collab" and set(o) >= {"cls_preds", "reg_preds", "dir_preds", "occ_single_list", "preds_tensor"}


def test_quant_twin_structure_and_parameters(qt):
    from quantv2x_amd.plugin.quant.quant_layer import QuantModule
    assert list(qt.state_dict().keys()) == [str(k) for k in G["quant_state_dict_keys"]]
    mods = dict(qt.model.named_modules())
    names = [n for n, m in mods.items() if isinstance(m, QuantModule)]
    assert names == [str(n) for n in G["module_names"]]
    blocks = [n for n, m in mods.items() if type(m).__name__ in ("QuantBasicBlock", "QuantBottleneck")]
    assert blocks == [str(n) for n in G["block_names"]]
    for n in names:
        m, key = mods[n], n.replace(".", "/")
        np.testing.assert_allclose(m.weight_quantizer.delta.reshape(-1).numpy(), G[key + "/w_delta"], rtol=1e-6)
        np.testing.assert_array_equal(m.weight_quantizer.zero_point.reshape(-1).numpy(), G[key + "/w_zp"])
        assert bool(m.disable_act_quant) == bool(G[key + "/a_off"]), n
        if not m.disable_act_quant:
            np.testing.assert_allclose(float(m.act_quantizer.delta), float(G[key + "/a_delta"]), rtol=1e-5, err_msg=n)
            assert float(m.act_quantizer.zero_point) == float(G[key + "/a_zp"]), n
    for n in blocks:
        key = n.replace(".", "/")
        np.testing.assert_allclose(float(mods[n].act_quantizer.delta), float(G[key + "/a_delta"]), rtol=1e-5, err_msg=n)
        assert float(mods[n].act_quantizer.zero_point) == float(G[key + "/a_zp"]), n


def test_quant_twin_activations_and_predictions(qt):
    from quantv2x_amd.plugin.quant.quant_layer import QuantModule
    model, outs, hooks = qt.model, {}, []
    for name, m in model.named_modules():
        if isinstance(m, QuantModule) or type(m).__name__ in ("QuantBasicBlock", "QuantBottleneck"):
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: outs.__setitem__(name, o)))
    dd = scene(2)
    with torch.no_grad():
        codes, _, info = model.encode_features(dd)
        o = model.decode_features(codes, info)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        n1 = model.forward_with_encdec(scene(1))
        n3 = model.forward_with_encdec(scene(3))
    np.testing.assert_array_equal(torch.stack([c[:, 0] for c in codes]).numpy().astype(np.uint8), G["w8a8/codes"])
    mods, checked = dict(model.named_modules()), 0
    for name, t in outs.items():
        key = name.replace(".", "/")
        if key + "/out_code" in G.files:
            aq = mods[name].act_quantizer
            code = torch.round(t / aq.delta + aq.zero_point).numpy().astype(np.uint8)
            bad = (code != G[key + "/out_code"]).mean()
            assert bad < 2e-3, (name, bad)          # fp32 conv re-association across torch builds: rare +-1 flips at most
            checked += 1
        if key + "/out_f32" in G.files:
            np.testing.assert_allclose(t.numpy()[:, ::4], G[key + "/out_f32"], rtol=1e-4, atol=1e-4, err_msg=name)
    assert checked >= 19 + 3 + 3
    lsb = max(float(mods[h].act_quantizer.delta) for h in ("cls_head", "reg_head", "dir_head"))
    for got, key in ((o, "w8a8/preds_tensor"), (n1, "w8a8/preds_tensor_n1"), (n3, "w8a8/preds_tensor_n3")):
        d = np.abs(got["preds_tensor"].numpy() - G[key])
        assert d.max() <= 2.001 * lsb and (d > 1e-5).mean() < 5e-3, (key, d.max(), (d > 1e-5).mean())


def test_export_refuses_what_the_engine_does_not_build(qt):
    from quantv2x_amd.ptq_state import export_ptq_state
    st = export_ptq_state(qt)
    keys = sorted(k for k in st if not k.startswith("meta/"))
    assert keys == [str(k) for k in G["ptq_export/keys"]]
    got = np.array([float(np.asarray(st[k], dtype=np.float64).sum()) for k in keys])
    np.testing.assert_allclose(got, G["ptq_export/checksum"], rtol=1e-5, atol=1e-4)
    assert str(st["meta/fusion_method"]) == "pyramid" and len(st["meta/block_names"]) == 19
    qt.model.pyramid_backbone.stage = "single"
    try:
        with pytest.raises(NotImplementedError):
            export_ptq_state(qt)
    finally:
        qt.model.pyramid_backbone.stage = "collab"
