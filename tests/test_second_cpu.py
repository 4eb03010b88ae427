"""a13 on the CPU: the sparse-convolution mirror against dense ``F.conv3d``, the QuantModel surgery on a SECOND encoder, the integer
restatement (oracle/spec_second.py) against the fake-quant mirror, and the C ABI's argument checks (no launches)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _common_second import build_second, calibrated_second, second_scene_np


@pytest.mark.parametrize("cls,kw", [("SubMConv3d", dict(kernel_size=3, padding=1)), ("SparseConv3d", dict(kernel_size=3, stride=2, padding=1)),
                                    ("SparseConv3d", dict(kernel_size=3, stride=2, padding=(0, 1, 1))),
                                    ("SparseConv3d", dict(kernel_size=(3, 1, 1), stride=(2, 1, 1), padding=0))])
def test_sparse_convolution_is_dense_convolution_on_the_active_set(cls, kw):
    """The published spconv semantics this build restates: F.conv3d on the densified volume, kept at the active outputs."""
    from quantv2x_amd.plugin.models.sub_modules import sparse_ops as S
    torch.manual_seed(0)
    b, d, h, w, c = 2, 9, 16, 24, 5
    occ = torch.rand(b, d, h, w) < 0.1
    idx = occ.nonzero().int()
    x = S.SparseConvTensor(torch.randn(idx.shape[0], c), idx, [d, h, w], b)
    m = getattr(S, cls)(c, 7, **kw)
    y = m(x)
    ref = F.conv3d(x.dense(), m.weight.permute(0, 4, 1, 2, 3), stride=m.stride, padding=m.padding)
    act = occ if m.subm else F.conv3d(occ[:, None].float(), torch.ones(1, 1, *m.kernel_size), stride=m.stride, padding=m.padding)[:, 0] > 0
    got_act = torch.zeros_like(act)
    i = y.indices.long()
    got_act[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = True
    assert torch.equal(got_act, act)
    assert (y.dense() - ref * act[:, None]).abs().max() < 1e-5
    assert list(ref.shape[2:]) == y.spatial_shape


def test_quant_model_wraps_second():
    qm = calibrated_second()
    enc = qm.model.encoder_m1
    assert type(enc).__name__ == "QuantSECOND" and type(enc.spconv_block).__name__ == "QuantVoxelBackBone8x"
    from quantv2x_amd.ptq_state import second_layers
    names = [n for n, _ in second_layers(enc)]
    assert names[0] == "spconv_block.conv_input.quant_conv_0" and names[2] == "spconv_block.conv2.layer_0.quant_conv_0" and len(names) == 12
    for _, m in second_layers(enc):
        assert type(m.norm_function).__name__ == "BatchNorm1d" and type(m.activation_function).__name__ == "ReLU"    # BN1d is not folded
        assert m.act_quantizer.inited and float(m.act_quantizer.zero_point) == 0.0                                      # post-ReLU ranges start at 0


def test_fake_quant_stays_close_to_fp32():
    enc = build_second()
    sc = second_scene_np()
    dd = {"inputs_m1": {k: torch.from_numpy(v) for k, v in sc.items()}}
    with torch.no_grad():
        ref = enc(dd, "m1")
        got = calibrated_second()(dd)
    assert ref.shape == (2, 256, 16, 32)
    assert (got - ref).abs().max() < 0.1 * ref.abs().max()


@pytest.mark.parametrize("num_features_out", [128, 64])
def test_oracle_agrees_with_the_mirror(num_features_out):
    """Whole path, not teacher-forced: a rounding flip in an early layer propagates through up to eleven more, so the end result is
    compared loosely here (a few codes on ~1 % of the cells); the layer-by-layer pin is the teacher-forced test below."""
    from oracle.spec_second import OracleSecond
    from quantv2x_amd.ptq_state import export_second_state
    qm = calibrated_second(num_features_out=num_features_out)
    sc = second_scene_np()
    with torch.no_grad():
        ref = qm({"inputs_m1": {k: torch.from_numpy(v) for k, v in sc.items()}}).numpy()
    state = export_second_state(qm.model.encoder_m1)
    orc = OracleSecond(state)
    got = orc.dequant(orc.forward(sc))
    lsb = float(state["second/11/a_delta"])
    d = np.abs(got - ref)
    assert got.shape == ref.shape == (2, 2 * num_features_out, 16, 32)
    assert d.max() <= 6.001 * lsb and (d > 1e-4).mean() < 2e-2, (d.max() / lsb, (d > 1e-4).mean())


def test_oracle_layers_teacher_forced():
    """Each layer on the MIRROR's own input codes: codes equal up to one step where a value sits on a rounding boundary."""
    from oracle.spec_second import OracleSecond, mean_vfe
    from quantv2x_amd.plugin.models.sub_modules.sparse_ops import SparseConvTensor
    from quantv2x_amd.ptq_state import export_second_state, second_layers
    qm = calibrated_second()
    enc = qm.model.encoder_m1
    state = export_second_state(enc)
    orc = OracleSecond(state)
    sc = second_scene_np()
    feats = mean_vfe(sc["voxel_features"], sc["voxel_num_points"])
    x = SparseConvTensor(torch.from_numpy(feats), torch.from_numpy(sc["voxel_coords"]), orc.shape0, 2)
    codes = feats
    for i, (_, m) in enumerate(second_layers(enc)):
        with torch.no_grad():
            y = m(x)
        got, oidx, osh = orc.layer(i, codes, x.indices.numpy(), x.spatial_shape)
        assert list(osh) == y.spatial_shape and np.array_equal(oidx, y.indices.numpy())
        want = np.rint(y.features.numpy() / float(state[f"second/{i}/a_delta"]) + float(state[f"second/{i}/a_zp"]))
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (i, d.max(), (d > 0).mean())
        x, codes = y, want.astype(np.uint8)


def test_export_refuses_unfrozen_or_non_w8a8():
    from quantv2x_amd.ptq_state import export_second_state, second_layers
    qm = calibrated_second()
    enc = qm.model.encoder_m1
    second_layers(enc)[3][1].act_quantizer.set_inited(False)
    with pytest.raises(ValueError):
        export_second_state(enc)
    second_layers(enc)[3][1].act_quantizer.set_inited(True)
    second_layers(enc)[5][1].act_quantizer.bitwidth_refactor(4)        # sub-8-bit ACTIVATIONS: the epilogues clamp to [0, 255]
    with pytest.raises(ValueError):
        export_second_state(enc)
    second_layers(enc)[5][1].act_quantizer.bitwidth_refactor(8)
    second_layers(enc)[5][1].weight_quantizer.bitwidth_refactor(4)     # sub-8-bit weights export (WxA8, round 3)
    assert "second/5/w_code" in export_second_state(enc)


def test_cabi_argument_checks():
    from quantv2x_amd import lib as L
    lib = L.load()
    d = L.SpconvDesc()
    d.subm = 0
    for a in range(3):
        d.k[a], d.s[a], d.p[a], d.in_shape[a], d.out_shape[a] = 3, 2, 1, 41, 21
    d.agents, d.cin, d.cout, d.cap_in, d.cap_out, d.out_delta, d.out_zp = 1, 32, 32, 64, 64, 0.1, 0.0
    one = C.c_void_p(16)
    assert lib.qv2x_sp_out_sites(C.byref(d), None, one, one, one, one, one, 1 << 20, None) == -1
    assert lib.qv2x_sp_out_sites_workspace_bytes(C.byref(d)) == (21 * 21 * 21 + 1023) // 1024 * 4 + 16
    assert lib.qv2x_sp_out_sites(C.byref(d), one, one, one, one, one, one, 8, None) == -1 and b"workspace" in lib.qv2x_last_error()
    d.out_shape[1] = 20
    assert lib.qv2x_sp_rulebook(C.byref(d), one, one, one, one, None) == -1 and b"out_shape" in lib.qv2x_last_error()
    d.out_shape[1] = 21
    d.cin = 48
    assert lib.qv2x_sp_conv_i8(C.byref(d), one, one, one, one, one, one, one, one, one, one, None) == -2
    d.subm = 1
    assert lib.qv2x_sp_out_sites(C.byref(d), one, one, one, one, one, one, 1 << 20, None) == -1
    assert lib.qv2x_mean_vfe_f32(one, one, one, 0, 5, one, None) == -1
    assert lib.qv2x_sp_to_bev_i8(one, one, one, 8, 128, 128, 1, 2, 8, 16, 300, one, None) == -1


def test_whole_model_with_a_second_modality():
    """``core_method: second`` in front of the same backbone / shrinker / codebook / fusion: QuantModel wraps it, the export carries
    both halves, and the integer restatement tracks the mirror (the codebook argmin is discontinuous, so agreement past it is counted)."""
    from _common import hard_forward
    from _common_second import calibrated_second_model, second_model_scene_np
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.ptq_state import export_ptq_state
    qt = calibrated_second_model()
    assert type(qt.model.encoder_m1).__name__ == "QuantSECOND" and qt.model.backbone_m1.blocks[0][1].weight.shape[1] == 256
    state = export_ptq_state(qt)
    assert str(state["meta/encoder"]) == "second" and list(state["meta/grid"]) == [32, 16, 2] and int(state["meta/canvas_channels"]) == 256
    assert list(state["meta/layer_strides"]) == [1, 2, 2] and "pfn/a2_delta" not in state
    sc = second_model_scene_np()
    ot, mt = {}, {}
    Oracle(state).forward(sc, ot)
    with torch.no_grad():
        hard_forward(qt.model, synth.scene_to_torch(sc), mt)
    lsb, zp = float(state["second/11/a_delta"]), float(state["second/11/a_zp"])
    canvas = (ot["canvas"].astype(np.float32) - zp) * lsb
    d = np.abs(canvas.transpose(0, 3, 1, 2) - mt["spatial_features"].numpy())
    assert d.max() <= 2.001 * lsb and (d > 1e-4).mean() < 1e-2
    assert (ot["codes"] == mt["codes"].numpy()).mean() > 0.85
