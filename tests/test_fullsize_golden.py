"""full-size checksums.  ``tests/golden/fullsize.npz`` holds per-tensor [sum, sum|x|, absmax] of the REFERENCE's
fp32 and W8A8 (min-max, frozen after one pass) forward at V2X-Real and OPV2V shape, every quantizer's (delta, zero point), the checksum
of every layer's integer weight codes and the histogram of every QuantModule's output codes -- tensors of 36-67 MB each are too large to
commit.  Checked here, on the CPU: the torch mirror reproduces them (so the PTQ state the full-size GPU tests deploy IS the reference's),
and the integer path of the CPU oracle lands on the reference's code histograms layer by layer."""
import numpy as np
import pytest
import torch

from _common import build_plugin, calibrated_plugin, hard_forward, scene

SHAPES = ["v2xreal", "opv2v"]


def _stats(t):
    a = t.detach().double()
    return np.array([a.sum().item(), a.abs().sum().item(), a.abs().max().item()])


def _weight_checksums(code):
    c = code.reshape(-1).astype(np.int64)
    return np.array([c.sum(), (c * (1 + np.arange(c.size) % 251)).sum()], dtype=np.int64)


@pytest.mark.parametrize("shape", SHAPES)
def test_mirror_fp32_forward_at_full_size(golden, shape):
    g = golden["fullsize"]
    torch.set_num_threads(8)
    taps = {}
    with torch.no_grad():
        hard_forward(build_plugin(shape), scene(1, shape, n_points=60000), taps)
    for k in ("spatial_features", "backbone", "shrinker", "decoded", "fused", "preds_tensor"):
        # sums of 1e7 positive fp32 terms: the order of a multi-threaded convolution moves them in the 6th digit at most
        np.testing.assert_allclose(_stats(taps[k]), g[f"{shape}/fp32/{k}"], rtol=2e-5, err_msg=k)
    hist = np.stack([np.bincount(taps["codes"][l].reshape(-1).numpy().astype(np.int64), minlength=128) for l in range(3)])
    assert np.abs(hist - g[f"{shape}/fp32/code_hist"]).sum() <= 2e-3 * hist.sum()      # indices: equal up to fragile rows


@pytest.mark.parametrize("shape", SHAPES)
def test_mirror_w8a8_state_and_oracle_codes_at_full_size(golden, shape):
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.ptq_state import export_ptq_state
    g = golden["fullsize"]
    torch.set_num_threads(8)
    qt = calibrated_plugin(shape, n_agents=1, n_points=60000)
    names = [str(n) for n in g[f"{shape}/w8a8/module_names"]]
    mods = dict(qt.model.named_modules())
    assert [n for n, m in qt.model.named_modules() if hasattr(m, "weight_quantizer") and hasattr(m, "org_weight")] == names
    for name in names:
        m, key = mods[name], f"{shape}/w8a8/" + name.replace(".", "/")
        wq, aq = m.weight_quantizer, m.act_quantizer
        code = torch.clamp(torch.round(m.weight / wq.delta) + wq.zero_point, 0, 255).detach().numpy().astype(np.uint8)
        np.testing.assert_array_equal(_weight_checksums(code), g[key + "/w_code_checksum"], err_msg=name)
        np.testing.assert_allclose(float(wq.delta.double().sum()), float(g[key + "/w_delta_sum"]), rtol=1e-6, err_msg=name)
        # activation ranges come out of a full-size fp32 forward: min / max of 1e7 values whose last bits depend on the summation order
        np.testing.assert_allclose(float(aq.delta), float(g[key + "/a_delta"]), rtol=1e-5, err_msg=name)
        assert float(aq.zero_point) == float(g[key + "/a_zp"]), name
    # the integer path (CPU oracle; the HIP kernels are bit-identical to it) against the reference's fake-quant codes, layer by layer
    state = export_ptq_state(qt)
    sc = synth.make_scene(shape, n_agents=1, seed=3, n_points=60000)
    taps = {}
    Oracle(state).forward(sc, taps)
    worst = 0.0
    for name in names:
        if name not in taps or not (name.startswith("backbone_m1.blocks") or name.startswith("shrinker_m1")):
            continue
        hist = np.bincount(taps[name].reshape(-1).astype(np.int64), minlength=256)
        ref = g[f"{shape}/w8a8/" + name.replace(".", "/") + "/out_hist"]
        assert hist.sum() == ref.sum(), name
        # a +-1 flip moves one count between neighbouring bins; flips compound over the 21 layers of this random-weight stack
        frac = np.abs(hist - ref).sum() / (2.0 * hist.sum())
        worst = max(worst, frac)
        assert frac < 0.02, (name, frac)
    codes = taps["codes"].reshape(3, -1)
    hist = np.stack([np.bincount(codes[l].astype(np.int64), minlength=128) for l in range(3)])
    assert np.abs(hist - g[f"{shape}/w8a8/code_hist"]).sum() / (2.0 * hist.sum()) < 0.05
    print(shape, "largest per-layer histogram distance (fraction of the codes):", worst)
