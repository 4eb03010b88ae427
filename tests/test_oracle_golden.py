"""The CPU oracle (``oracle/``) against golden vectors captured from the reference: this is what pins it."""
import numpy as np
import pytest
import torch

from _common import calibrated_plugin, scene_np, sub8

from oracle import geometry
from oracle.spec import Oracle
from quantv2x_amd.ptq_state import export_ptq_state

torch.set_num_threads(1)


@pytest.fixture(scope="module")
def state():
    return export_ptq_state(calibrated_plugin())


def _weight_checksums(code):
    c = code.reshape(-1).astype(np.int64)
    return np.array([c.sum(), (c * (1 + np.arange(c.size) % 251)).sum()], dtype=np.int64)


def test_exported_state_matches_reference_quantizers(golden, state):
    g = golden["tiny_w8a8"]
    assert [str(n) for n in state["meta/module_names"]] == [str(n) for n in g["module_names"]]
    for n in g["module_names"]:
        k = str(n).replace(".", "/")
        np.testing.assert_array_equal(state[f"{n}/w_delta"], g[k + "/w_delta"])
        np.testing.assert_array_equal(state[f"{n}/w_zp"], g[k + "/w_zp"])
        np.testing.assert_array_equal(_weight_checksums(state[f"{n}/w_code"]), g[k + "/w_code_checksum"])
        np.testing.assert_allclose(state[f"{n}/a_delta"], g[k + "/a_delta"], rtol=1e-6)
        assert float(state[f"{n}/a_zp"]) == float(g[k + "/a_zp"])


def _nhwc(g, name):
    return np.ascontiguousarray(g[name.replace(".", "/") + "/out_code"].transpose(0, 2, 3, 1))


def test_integer_path_vs_reference_fake_quant(golden, state):
    """Every uint8 activation the reference's fake-quant forward produced, reproduced layer by layer by the
    integer restatement from the reference's own layer input: identical except for <= 1 LSB on a tiny fraction
    (the reference sums its fp32 convolutions in an unspecified order; the restatement sums exactly).
    Layers are teacher-forced because a single +-1 flip is amplified chaotically by the random-weight
    19-layer stack -- that is a property of the network, not of either implementation."""
    g = golden["tiny_w8a8"]
    orc = Oracle(state)
    sc = scene_np(2)
    pcodes, canvas, cq = orc.pfn_scatter(sc, 2)
    np.testing.assert_array_equal(pcodes, g["pfn/pillar_code"])      # PFN + both quantizers: bit exact
    flips, total = 0, 0

    def check(name, got):
        nonlocal flips, total
        diff = got.astype(np.int32) - _nhwc(g, name).astype(np.int32)
        assert np.abs(diff).max() <= 1, name
        flips += int((diff != 0).sum()); total += diff.size

    x, xq = canvas, cq
    cat_parts, cat_q, c0 = [], [], 0
    for lvl in range(3):
        for i in range(orc.layer_nums[lvl] + 1):
            name = f"backbone_m1.blocks.{lvl}.{i + 1}"
            got, xq = orc.conv(name, x, [(0, x.shape[3], xq[0], xq[1])], stride=orc.strides[lvl] if i == 0 else 1)
            check(name, got)
            x = _nhwc(g, name)                                        # teacher forcing
        name = f"backbone_m1.deblocks.{lvl}.0"
        s_up = orc.ups[lvl]
        up = np.zeros((x.shape[0], x.shape[1] * s_up, x.shape[2] * s_up, 128), np.uint8)
        oq = orc.deconv(name, x, xq, s_up, up, 0)
        check(name, up)
        cat_parts.append(_nhwc(g, name)); cat_q.append((c0, 128, oq[0], oq[1])); c0 += 128
    cat = np.concatenate(cat_parts, axis=-1)
    got, q0 = orc.conv("shrinker_m1.layers.0.double_conv.0", cat, cat_q)
    check("shrinker_m1.layers.0.double_conv.0", got)
    x = _nhwc(g, "shrinker_m1.layers.0.double_conv.0")
    got, q1 = orc.conv("shrinker_m1.layers.0.double_conv.1", x, [(0, 256, q0[0], q0[1])])
    check("shrinker_m1.layers.0.double_conv.1", got)
    assert flips / total < 5e-4, (flips, total)

    shr = _nhwc(g, "shrinker_m1.layers.0.double_conv.1")
    codes = orc.encode(shr, q1)
    ref_codes = g["hard/codes"].reshape(3, -1)
    assert (codes != ref_codes).mean() < 2e-3
    feats = orc.decode(ref_codes).reshape(2, shr.shape[1], shr.shape[2], 256)
    np.testing.assert_allclose(sub8(feats.transpose(0, 3, 1, 2)), g["hard/decoded"], rtol=1e-4, atol=2e-5)
    fused = orc.fuse(feats, sc["pairwise_t_matrix"], sc["record_len"])
    np.testing.assert_allclose(sub8(fused.transpose(0, 3, 1, 2)), g["hard/fused"], rtol=1e-4, atol=2e-5)
    preds = np.concatenate(orc.heads(fused), axis=1)
    d = np.abs(preds - g["hard/preds_tensor"])
    lsb = max(float(state[h + "/a_delta"]) for h in ("cls_head", "reg_head", "dir_head"))
    assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 2e-3


def test_end_to_end_chain_runs_and_is_deterministic(state):
    a = Oracle(state).forward(scene_np(2))
    b = Oracle(state).forward(scene_np(2))
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["preds_tensor"].shape == (1, 72, 16, 32) and a["cls_preds_single"].shape == (2, 2, 16, 32)


def test_codebook_encode_indices_bit_exact_vs_reference(golden, state):
    g = golden["codebook"]
    codes, gaps = Oracle(state).encode_rows(g["x"], want_gaps=True)
    mism = codes != g["codes"]
    # rows whose top-2 distance gap is far above fp32 summation noise must agree exactly
    solid = g["gaps"] > 1e-4
    assert not (mism & solid).any()
    assert mism.mean() < 2e-3
    np.testing.assert_allclose(gaps[~mism], g["gaps"][~mism], rtol=0, atol=2e-4)


def test_decode_tables_equal_layered_decode(golden, state):
    g = golden["codebook"]
    dec = Oracle(state).decode(g["codes"])
    np.testing.assert_allclose(dec, g["decoded"], rtol=1e-5, atol=2e-6)


def test_geometry_restatement(golden):
    g = golden["geometry"]
    aff = geometry.normalize_pairwise_tfm(g["pairwise"], 12.8, 25.6, 1)
    np.testing.assert_array_equal(aff, g["affine_f64"])
    src = g["src"].transpose(0, 2, 3, 1)
    warped = geometry.warp_to_ego(src, aff[0], 4)
    np.testing.assert_allclose(warped.transpose(0, 3, 1, 2), g["warped"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(geometry.att_fuse(warped).transpose(2, 0, 1), g["att_fused"][0], rtol=1e-5, atol=2e-6)
    one = geometry.att_fuse(geometry.warp_to_ego(src[:1], aff[0], 1))
    np.testing.assert_allclose(one.transpose(2, 0, 1), g["att_fused_n1"][0], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("n", [1, 3])
def test_other_agent_counts_shapes(state, n):
    out = Oracle(state).forward(scene_np(n))
    assert out["preds_tensor"].shape == (1, 72, 16, 32) and np.isfinite(out["preds_tensor"]).all()
    assert out["cls_preds_single"].shape[0] == n


def test_freeze_step_gives_the_same_state_on_the_reference_objects(golden, state):
    """``export_ptq_state`` was run on the reference's own ``QuantModel`` when the goldens were made (it only reads
    attribute names): same keys, same contents as on the mirror -- ``quantv2x_amd.deploy`` accepts the reference's objects."""
    g = golden["tiny_w8a8"]
    keys = [str(k) for k in g["ptq_export/keys"]]
    # meta/fusion_method, meta/compress and meta/encoder are newer than the golden file (round 2: the export names the fusion / compressor / encoder it found)
    assert keys == sorted(k for k in state if k not in ("meta/module_names", "meta/fusion_method", "meta/compress", "meta/encoder", "meta/w_bits", "meta/codebook_segs"))   # (round 5: seg_num)
    got_sum = np.array([float(np.asarray(state[k], dtype=np.float64).sum()) for k in keys])
    got_abs = np.array([float(np.abs(np.asarray(state[k], dtype=np.float64)).sum()) for k in keys])
    np.testing.assert_allclose(got_sum, g["ptq_export/checksum"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(got_abs, g["ptq_export/absum"], rtol=1e-6, atol=1e-9)
