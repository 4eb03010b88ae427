"""Heterogeneous scenes (heter_model_baseline.py:41-75, 169-216: one encoder / backbone / shrinker per modality, features assembled in
``agent_modality_list`` order, shared codebook / fusion / heads) against ``tests/golden/tiny_heter_w8a8.npz`` -- the reference's own
two-modality QuantModel on the [m1, m2, m1] scene (``make_golden.py heter``).  CPU: the mirror and the per-modality PTQ export + oracle;
``-m gpu``: ``DeployedHeterModel`` (one engine per modality, the code planes put in agent order) against the oracle, bit for bit."""
import numpy as np
import pytest
import torch

from _common import HETER_MODALITIES, calibrated_heter_plugin, hard_forward_heter, head_lsb, heter_oracle_forward, heter_scene_np, sub8
from quantv2x_amd import synth


import contextlib


@contextlib.contextmanager
def one_thread():
    """As make_golden.py runs: the m2 stack sees a batch of ONE agent, where torch's CPU convolution sums in a thread-count dependent
    order -- a last-bit difference that one rounding flip turns into a 5e-4 shift of a later min-max range, or into another code."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        yield
    finally:
        torch.set_num_threads(n)


@pytest.fixture(scope="module")
def qt():
    with one_thread():
        return calibrated_heter_plugin()


@pytest.fixture(scope="module")
def states(qt):
    from quantv2x_amd.ptq_state import export_ptq_state
    return {m: export_ptq_state(qt, modality=m) for m in ("m1", "m2")}


def test_mirror_matches_the_reference_on_a_two_modality_model(golden, qt):
    from quantv2x_amd.plugin.quant import QuantModule
    g = golden["tiny_heter_w8a8"]
    model = qt.model
    assert model.modality_name_list == ["m1", "m2"]
    assert sorted(model.state_dict().keys()) == list(g["state_dict_keys"])
    assert [n for n, m in model.named_modules() if isinstance(m, QuantModule)] == list(g["module_names"])
    for n, m in model.named_modules():
        if not isinstance(m, QuantModule):
            continue
        k = n.replace('.', '/')
        wq, aq = m.weight_quantizer, m.act_quantizer
        np.testing.assert_array_equal(wq.delta.detach().numpy().reshape(-1), g[k + '/w_delta'])
        np.testing.assert_array_equal(wq.zero_point.detach().numpy().reshape(-1), g[k + '/w_zp'])
        np.testing.assert_allclose(np.float32(aq.delta), g[k + '/a_delta'], rtol=1e-6, err_msg=n)
        assert float(aq.zero_point) == float(g[k + '/a_zp']), n
    taps = {}
    with torch.no_grad(), one_thread():
        hard_forward_heter(model, synth.scene_to_torch(heter_scene_np()), taps)
    mism = (taps['codes'].numpy().astype(np.uint8) != g['hard/codes']).mean()
    assert mism < 5e-3
    if mism == 0:
        np.testing.assert_allclose(taps['preds_tensor'].numpy(), g['hard/preds_tensor'], rtol=1e-4, atol=1e-4)


def test_export_per_modality(states, qt):
    s1, s2 = states["m1"], states["m2"]
    assert sorted(set(s1) ^ set(s2)) == ["meta/modality"] and str(s2["meta/modality"]) == "m2"
    k = "backbone_m1.blocks.0.1/w_code"
    assert (s1[k] != s2[k]).mean() > 0.5                                   # each modality its own weights, under the canonical names
    w2 = qt.model.backbone_m2.blocks[0][1]
    code = torch.clamp(torch.round(w2.weight / w2.weight_quantizer.delta) + w2.weight_quantizer.zero_point, 0, 255).detach().numpy().astype(np.uint8)
    np.testing.assert_array_equal(s2[k], code)
    np.testing.assert_array_equal(s1["codebook/1/codebook"], s2["codebook/1/codebook"])      # shared parts are the same arrays
    np.testing.assert_array_equal(s1["cls_head/w_code"], s2["cls_head/w_code"])
    from quantv2x_amd.ptq_state import export_ptq_state
    with pytest.raises(ValueError):
        export_ptq_state(qt, modality="m3")


def test_oracle_per_modality_tracks_the_reference(golden, states):
    """Free-running (not teacher-forced) through 2 x 22 layers of a random-weight model: a rounding flip early on moves later codes, so
    the integer restatement tracks the reference's fake-quant forward statistically (the layer-by-layer, teacher-forced pin of the same
    code path is tests/test_oracle_golden.py).  What this pins is the ASSEMBLY: every agent's codes come from ITS modality's stack, in
    agent order -- a swapped or mis-indexed agent would disagree on ~99 % of its cells, not on 3 %."""
    g = golden["tiny_heter_w8a8"]
    sc = heter_scene_np()
    taps = {}
    out = heter_oracle_forward(states, sc, taps)
    for m in ("m1", "m2"):                                                 # each modality's shrinker output codes, every 8th channel
        want = g[f"hard/shrinker_{m}_code"].transpose(0, 2, 3, 1)
        d = np.abs(taps["shrinker_" + m][..., ::8].astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 8 and (d != 0).mean() < 0.3, (m, d.max(), (d != 0).mean())
    for a in range(3):
        assert (taps["codes"][:, a] != g["hard/codes"][:, a]).mean() < 0.06, a
    swapped = taps["codes"][:, [1, 0, 2]]
    assert (swapped[:, 0] != g["hard/codes"][:, 0]).mean() > 0.9          # (the yardstick: another agent's codes)
    assert out["preds_tensor"].shape == g["hard/preds_tensor"].shape and out["cls_preds_single"].shape[0] == 3
    # a modality's agent through the composed path == the same agent through that modality's single-modality oracle
    from oracle.spec import Oracle
    one = {"inputs_m1": sc["inputs_m2"], "agent_modality_list": ["m1"], "record_len": np.asarray([1]), "pairwise_t_matrix": sc["pairwise_t_matrix"]}
    t1 = {}
    Oracle(states["m2"]).forward(one, t1)
    np.testing.assert_array_equal(t1["codes"][:, 0], taps["codes"][:, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("modalities", [HETER_MODALITIES, ["m2", "m1"], ["m2"], ["m1", "m1", "m2", "m2"]])
def test_deployed_heterogeneous_scene_bit_exact_vs_oracle(qt, states, modalities):
    from quantv2x_amd.engine import DeployedHeterModel, deploy
    eng = deploy(qt)
    assert isinstance(eng, DeployedHeterModel) and sorted(eng.engines) == ["m1", "m2"]
    sc = heter_scene_np(modalities)
    otaps, gtaps = {}, {}
    want = heter_oracle_forward(states, sc, otaps)
    got = eng(synth.scene_to_torch(sc, "cuda"), gtaps)
    torch.cuda.synchronize()
    codes = gtaps["codes"].cpu().numpy().reshape(otaps["codes"].shape)
    np.testing.assert_array_equal(codes, otaps["codes"])                  # every agent's wire indices, in agent order
    d = np.abs(got["preds_tensor"].cpu().numpy() - want["preds_tensor"])
    lsb = head_lsb(states["m1"])
    assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 1e-3
    ds = np.abs(got["cls_preds_single"].cpu().numpy() - want["cls_preds_single"])
    assert ds.max() <= head_lsb(states["m1"], "_single") * 1.001 and (ds > 1e-5).mean() < 1e-3


@pytest.mark.gpu
def test_heterogeneous_frame_replays_as_one_hip_graph(qt):
    from quantv2x_amd.engine import deploy
    eng = deploy(qt)
    dd = synth.scene_to_torch(heter_scene_np(), "cuda")
    eager = {k: v.clone() for k, v in eng(dd).items()}
    replay = eng.capture(dd)
    for _ in range(3):
        out = replay()
    torch.cuda.synchronize()
    for k in eager:
        assert torch.equal(out[k], eager[k]), k


@pytest.mark.gpu
def test_single_modality_engine_still_refuses_other_modalities(states):
    from quantv2x_amd.engine import deploy
    eng = deploy(state=states["m1"])
    with pytest.raises(NotImplementedError):
        eng(synth.scene_to_torch(heter_scene_np(), "cuda"))


# ---- MIXED encoders: m1 = PointPillar, m3 = SECOND in one scene (heter_model_baseline.py:47-59, heter_encoders.py:52-81) ----
@pytest.fixture(scope="module")
def mixed():
    from _common import calibrated_mixed_plugin
    from quantv2x_amd.ptq_state import export_ptq_state
    with one_thread():
        qt = calibrated_mixed_plugin()
    return qt, {m: export_ptq_state(qt, modality=m) for m in ("m1", "m3")}


def test_mixed_encoder_model_mirror_export_and_oracle(mixed):
    """The mirror builds the reference-shaped two-ENCODER model (PointPillar + SECOND by ``core_method``), QuantModel wraps both stacks,
    every modality exports its own state (the SECOND one with its sparse convolutions and ``meta/encoder``), and the composed oracle --
    each modality's agents through ITS encoder, code planes in agent order -- tracks the mirror's hard forward (spconv is absent from
    /root/reference: the SECOND half is pinned on the mirror, not on the reference)."""
    from _common import MIXED_MODALITIES, hard_forward_heter, heter_oracle_forward, mixed_scene_np
    qt, states = mixed
    model = qt.model
    assert model.modality_name_list == ["m1", "m3"]
    assert type(model.encoder_m1).__name__ == "QuantPointPillar" and type(model.encoder_m3).__name__ == "QuantSECOND"
    assert "meta/encoder" not in states["m1"] or str(states["m1"]["meta/encoder"]) != "second"
    assert str(states["m3"]["meta/encoder"]) == "second" and int(states["m3"]["meta/canvas_channels"]) == 256
    assert list(states["m3"]["meta/layer_strides"]) == [1, 2, 2] and list(states["m1"]["meta/layer_strides"]) == [2, 2, 2]
    np.testing.assert_array_equal(states["m1"]["codebook/1/codebook"], states["m3"]["codebook/1/codebook"])
    sc = mixed_scene_np()
    ot, mt = {}, {}
    out = heter_oracle_forward(states, sc, ot)
    with torch.no_grad(), one_thread():
        hard_forward_heter(model, synth.scene_to_torch(sc), mt)
    codes_m = mt["codes"].numpy().astype(np.uint8)
    for a, m in enumerate(MIXED_MODALITIES):                          # free-running through 22 random-weight layers: statistical agreement ...
        assert (ot["codes"][:, a] == codes_m[:, a]).mean() > 0.85, (a, m)
    assert (ot["codes"][:, [1, 0, 2]][:, 0] == codes_m[:, 0]).mean() < 0.2      # ... against the yardstick of another agent's codes
    assert out["preds_tensor"].shape == mt["preds_tensor"].shape
    # the SECOND agent through the composed path == the same agent through the single-modality SECOND oracle
    from oracle.spec import Oracle
    one = {"inputs_m1": sc["inputs_m3"], "agent_modality_list": ["m1"], "record_len": np.asarray([1]), "pairwise_t_matrix": sc["pairwise_t_matrix"]}
    t1 = {}
    Oracle(states["m3"]).forward(one, t1)
    np.testing.assert_array_equal(t1["codes"][:, 0], ot["codes"][:, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("modalities", [["m1", "m3", "m1"], ["m3", "m1"], ["m3"], ["m3", "m3", "m1", "m1"]])
def test_deployed_mixed_encoder_scene_bit_exact_vs_oracle(mixed, modalities):
    """PointPillar agents and SECOND agents in one deployed scene: every agent's wire indices equal to ITS modality's oracle, predictions
    within the head-LSB rule; then the same frame as ONE HIP graph."""
    from _common import heter_oracle_forward, mixed_scene_np
    from quantv2x_amd.engine import DeployedHeterModel, deploy
    qt, states = mixed
    eng = deploy(qt)
    assert isinstance(eng, DeployedHeterModel) and sorted(eng.engines) == ["m1", "m3"]
    assert eng.engines["m3"].encoder_kind == "second" and eng.engines["m1"].encoder_kind == "point_pillar"
    eng.engines["m3"].second_max_voxels = 4096
    sc = mixed_scene_np(modalities)
    otaps, gtaps = {}, {}
    want = heter_oracle_forward(states, sc, otaps)
    dd = synth.scene_to_torch(sc, "cuda")
    got = {k: v.clone() for k, v in eng(dd, gtaps).items()}
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gtaps["codes"].cpu().numpy().reshape(otaps["codes"].shape), otaps["codes"])
    d = np.abs(got["preds_tensor"].cpu().numpy() - want["preds_tensor"])
    assert d.max() <= head_lsb(states[modalities[0]]) * 1.001 and (d > 1e-5).mean() < 1e-3
    replay = eng.capture(dd)
    for _ in range(2):
        out = replay()
    torch.cuda.synchronize()
    for k in got:
        assert torch.equal(out[k], got[k]), k
