"""The host-side mirror of the reference's plugin API against golden vectors captured from the
reference itself (``tests/golden/make_golden.py``).  CPU only."""
import numpy as np
import pytest
import torch

from _common import (act_quantizers, build_plugin, calibrated_plugin, hard_forward, quant_wrap, scene, sub8)

torch.set_num_threads(1)
FP = dict(rtol=1e-4, atol=2e-5)   # fp32, same torch ops as the reference: thread/blocking noise only


def test_state_dict_keys_match_reference(golden):
    model = build_plugin()
    assert list(model.state_dict().keys()) == list(golden["tiny_fp32"]["state_dict_keys"])


def test_synthetic_inputs_unchanged(golden):
    dd = scene(2)
    got = [float(dd['inputs_m1'][k].double().sum()) for k in ('voxel_features', 'voxel_coords', 'voxel_num_points')]
    np.testing.assert_allclose(got, golden["tiny_fp32"]["in_checksum"], rtol=1e-12)


def test_fp32_stages(golden):
    g = golden["tiny_fp32"]
    model, dd = build_plugin(), scene(2)
    with torch.no_grad():
        bd = {k: dd['inputs_m1'][k] for k in dd['inputs_m1']}
        np.testing.assert_allclose(model.encoder_m1.pillar_vfe(dict(bd))['pillar_features'].numpy(), g['pillar_features'], **FP)
        x = model.encoder_m1(dd, 'm1')
        for lvl in range(3):
            x = model.backbone_m1.blocks[lvl](x)
            np.testing.assert_allclose(sub8(x), g[f'block{lvl}'], **FP)
            np.testing.assert_allclose(sub8(model.backbone_m1.deblocks[lvl](x)), g[f'up{lvl}'], **FP)
        taps = {}
        hard_forward(model, dd, taps)
    for k in ('backbone', 'shrinker'):
        np.testing.assert_allclose(sub8(taps[k]), g[k], **FP)
    np.testing.assert_allclose(taps['affine'].numpy(), g['affine'], rtol=1e-12)
    codes = taps['codes'].numpy().astype(np.uint8)
    mism = (codes != g['codes']).mean()
    assert mism <= 2e-3, f"codebook index mismatch rate {mism}"
    if mism == 0:
        np.testing.assert_allclose(sub8(taps['decoded']), g['decoded'], **FP)
        np.testing.assert_allclose(sub8(taps['fused']), g['fused'], **FP)
        np.testing.assert_allclose(taps['preds_tensor'].numpy(), g['preds_tensor'], **FP)


@pytest.mark.parametrize("n", [1, 3])
def test_fp32_other_agent_counts(golden, n):
    with torch.no_grad():
        got = hard_forward(build_plugin(), scene(n)).numpy()
    want = golden["tiny_fp32"][f'preds_tensor_n{n}']
    bad = np.abs(got - want) > (2e-5 + 1e-4 * np.abs(want))
    assert bad.mean() < 2e-3   # a flipped code changes a handful of cells


def test_soft_gumbel_forward_reproduces_reference_under_seed(golden):
    g = golden["tiny_fp32"]
    model, dd = build_plugin(), scene(2)
    with torch.no_grad():
        torch.manual_seed(0)
        out = model(dd)
    np.testing.assert_allclose(out['preds_tensor'].numpy(), g['soft_preds_tensor_seed0'], **FP)
    np.testing.assert_allclose(out['codebook_loss'].numpy(), g['soft_codebook_loss_seed0'], rtol=1e-5)
    np.testing.assert_allclose(out['cls_preds_single'].numpy(), g['cls_preds_single_soft_seed0'], **FP)


def test_single_class_no_codebook_model(golden):
    with torch.no_grad():
        out = build_plugin(multiclass=False, codebook=False)(scene(2))
    assert out['cls_preds'].shape[1] == 2 and out['reg_preds'].shape[1] == 14 and out['dir_preds'].shape[1] == 4
    np.testing.assert_allclose(out['preds_tensor'].numpy(), golden["tiny_fp32"]['plain_sc_preds_tensor'], **FP)


def test_hard_eval_switch_routes_through_encode_decode(golden):
    model = build_plugin()
    model.hard_eval = True
    with torch.no_grad():
        a = model(scene(2))['preds_tensor'].numpy()
        b = model(scene(2))['preds_tensor'].numpy()
    np.testing.assert_array_equal(a, b)   # deterministic, unlike the Gumbel path
    np.testing.assert_allclose(a, golden["tiny_fp32"]['preds_tensor'], **FP)


def test_create_model_miss_exits_zero():
    from quantv2x_amd.plugin.tools import train_utils
    with pytest.raises(SystemExit) as e:
        train_utils.create_model({'model': {'core_method': 'no_such_model', 'args': {}}})
    assert e.value.code == 0


# ------------------------------------------------------------------------------------- W8A8

def _weight_checksums(code):
    c = code.reshape(-1).astype(np.int64)
    return np.array([c.sum(), (c * (1 + np.arange(c.size) % 251)).sum()], dtype=np.int64)


def test_w8a8_qparams_weight_codes_and_activation_codes(golden):
    from quantv2x_amd.plugin.quant import QuantModule
    g = golden["tiny_w8a8"]
    qt = calibrated_plugin()
    model = qt.model
    names = [n for n, m in model.named_modules() if isinstance(m, QuantModule)]
    assert names == list(g['module_names'])
    assert len(act_quantizers(qt)) == 32
    outs, hooks = {}, []
    for n, m in model.named_modules():
        if isinstance(m, QuantModule):
            hooks.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, o)))
    dd = scene(2)
    with torch.no_grad():
        taps = {}
        hard_forward(model, dd, taps)
    for h in hooks:
        h.remove()
    for n, m in model.named_modules():
        if not isinstance(m, QuantModule):
            continue
        k = n.replace('.', '/')
        wq, aq = m.weight_quantizer, m.act_quantizer
        np.testing.assert_array_equal(wq.delta.detach().numpy().reshape(-1), g[k + '/w_delta'])
        np.testing.assert_array_equal(wq.zero_point.detach().numpy().reshape(-1), g[k + '/w_zp'])
        code = torch.clamp(torch.round(m.weight / wq.delta) + wq.zero_point, 0, 255).detach().numpy().astype(np.uint8)
        np.testing.assert_array_equal(_weight_checksums(code), g[k + '/w_code_checksum'])
        if (k + '/w_code') in g.files:
            np.testing.assert_array_equal(code, g[k + '/w_code'])
        np.testing.assert_allclose(np.float32(aq.delta), g[k + '/a_delta'], rtol=1e-6)
        assert float(aq.zero_point) == float(g[k + '/a_zp'])
        if (k + '/out_code') in g.files:
            got = torch.round(outs[n] / aq.delta + aq.zero_point).numpy().astype(np.uint8)
            diff = got.astype(np.int32) - g[k + '/out_code'].astype(np.int32)
            assert np.abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3, n
    pfn = model.encoder_m1.pillar_vfe.pfn_layers[0]
    np.testing.assert_allclose(np.float32(pfn.act_quantizer.delta), g['pfn/a2_delta'], rtol=1e-6)
    mism = (taps['codes'].numpy().astype(np.uint8) != g['hard/codes']).mean()
    assert mism < 5e-3
    if mism == 0:
        np.testing.assert_allclose(taps['preds_tensor'].numpy(), g['hard/preds_tensor'], rtol=1e-4, atol=1e-4)


def test_quant_state_toggle_and_output_disable():
    qt = quant_wrap(build_plugin())
    qt.set_quant_state(False, False)
    with torch.no_grad():
        a = hard_forward(qt.model, scene(2))
        ref = hard_forward(build_plugin(), scene(2))
    # BN folding only: same function up to fp32 re-association
    assert float((a - ref).abs().max()) < 1e-3
    qt.disable_network_output_quantization()
    assert qt.model.cls_head.disable_act_quant and qt.model.reg_head.disable_act_quant
    assert not qt.model.shrinker_m1.layers[0].double_conv[0].disable_act_quant
    assert "MB" in qt.get_memory_footprint()


def test_codebook_is_left_unquantized():
    from quantv2x_amd.plugin.quant import QuantModule
    qt = quant_wrap(build_plugin())
    assert not any(isinstance(m, QuantModule) for m in qt.model.codebook.modules())
    assert isinstance(qt.model.codebook._encoders[0]._latentStageEncoder, torch.nn.Linear)


def test_inference_wrappers_contract():
    """inference_intermediate_fusion(batch, model, dataset) -> {pred_box_tensor, pred_score, gt_box_tensor}; to_device recursion."""
    from quantv2x_amd.plugin.tools import inference_utils, train_utils

    class _Post:
        def generate_gt_bbx(self, batch):
            return "gt"

    class _Dataset:
        post_processor = _Post()

        def __init__(self, three):
            self.three = three

        def post_process(self, batch, out):
            assert set(out['ego']) >= {'cls_preds', 'reg_preds', 'dir_preds', 'preds_tensor'}
            return ("box", "score", "gt3") if self.three else ("box", "score")

    model = build_plugin()
    model.hard_eval = True
    batch = train_utils.to_device({'ego': scene(2), 'meta': [1, 'x', 2.0]}, 'cpu')
    assert batch['meta'] == [1, 'x', 2.0]
    with torch.no_grad():
        a = inference_utils.inference_intermediate_fusion(batch, model, _Dataset(True))
        b = inference_utils.inference_early_fusion(batch, model, _Dataset(False))
    assert a == {"pred_box_tensor": "box", "pred_score": "score", "gt_box_tensor": "gt3"}
    assert b["gt_box_tensor"] == "gt"
