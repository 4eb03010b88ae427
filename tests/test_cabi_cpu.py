"""CPU-side checks of the C ABI: the library loads, exports every symbol ``include/qv2x.h`` declares, rejects bad
arguments without touching a GPU, and the host packing helpers are right.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from quantv2x_amd import build, lib
    build.build()
    return lib.load()


def test_library_exports_every_declared_symbol():
    from quantv2x_amd import lib
    header = open(os.path.join(ROOT, "include", "qv2x.h")).read()
    declared = sorted(set(re.findall(r"\b(qv2x_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(lib.SYMBOLS)
    l = _lib()
    for s in declared:
        assert hasattr(l, s), s
    abi = int(re.search(r"#define QV2X_ABI_VERSION (\d+)", header).group(1))
    assert l.qv2x_version() == abi == lib.ABI_VERSION


def test_dynamic_symbol_table_is_the_header_only():
    """-fvisibility=hidden + the header's visibility pragma: `nm -D` shows the declared C entry points and nothing else of ours
    (no mangled qv2x:: helpers, no kernel stubs)."""
    import subprocess
    from quantv2x_amd import lib
    _lib()
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    names = [ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in ("T", "t", "W", "w", "B", "D", "V")]
    ours = [n for n in names if "qv2x" in n.lower()]
    assert sorted(ours) == sorted(lib.SYMBOLS), sorted(set(ours) ^ set(lib.SYMBOLS))
    leaked = [n for n in names if n.startswith("_ZN4qv2x") or "__device_stub__" in n]
    assert not leaked, leaked[:5]


def test_struct_layouts_match_header():
    from quantv2x_amd import lib
    assert C.sizeof(lib.PfnParams) == (640 + 64 + 4 + 6) * 4
    assert C.sizeof(lib.ConvDesc) == (7 + 3 * 4 + 3) * 4 + 8
    assert C.sizeof(lib.DeconvDesc) == 15 * 4
    assert C.sizeof(lib.EncodeDesc) == 8 * 4
    assert C.sizeof(lib.FuseDesc) == 7 * 4 + 4 + 2 * 8 + 3 * 8 + 4 + 4    # 4 bytes of padding before the int64 fields, `fusion` + tail padding
    assert C.sizeof(lib.Conv1x1Desc) == 14 * 4 and C.sizeof(lib.GconvDesc) == 9 * 4 and C.sizeof(lib.OccDesc) == 10 * 4


def test_argument_errors_without_gpu():
    from quantv2x_amd import lib
    l = _lib()
    assert l.qv2x_fill_i8(None, 16, 0, None) == -1
    assert b"null" in l.qv2x_last_error()
    d = lib.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc = 1, 4, 4, 3, 100
    buf = np.zeros(64, np.uint8)
    ptrs = (C.c_void_p * 3)()
    assert l.qv2x_codebook_encode_f32(C.byref(d), lib.ptr(buf), ptrs, lib.ptr(buf), None) == -1
    assert b"dict_size" in l.qv2x_last_error()
    assert l.qv2x_heads_f32(lib.ptr(buf), 10, 3, 72, 96, lib.ptr(buf), lib.ptr(buf), lib.ptr(buf), lib.ptr(buf), lib.ptr(buf), None) == -1
    # the workgroup form's section (three heads + biases, the codebook twice, |C|^2) + the wave form's (the four matrices again + padding)
    assert l.qv2x_codebook_level_floats(128) == (3 * (65536 + 256) + 256 * 128 * 2 + 128) + (3 * 65536 + 128 * 256 + 4096)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from quantv2x_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.Qv2xError):
        lib.load()


def test_deploy_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from _common import calibrated_plugin
    from quantv2x_amd import lib
    from quantv2x_amd.engine import deploy
    with pytest.raises(lib.Qv2xError):
        deploy(calibrated_plugin())


def test_pack_k4_and_decode_tables():
    from quantv2x_amd.engine import _pack_k4, decode_tables
    w = np.arange(6 * 8, dtype=np.float32).reshape(6, 8)
    p = _pack_k4(w)
    assert p.shape == (2, 6, 4)
    for j in range(6):
        for k in range(8):
            assert p[k // 4, j, k % 4] == w[j, k]
    # decode tables against the layered decode of the plugin codebook
    import torch
    from _common import build_plugin
    from quantv2x_amd.ptq_state import export_ptq_state
    from _common import calibrated_plugin
    qt = calibrated_plugin()
    state = export_ptq_state(qt)
    lut, bias = decode_tables(state, 3)
    cb = qt.model.codebook
    g = np.random.default_rng(0)
    codes = [torch.from_numpy(g.integers(0, 128, (50, 1))) for _ in range(3)]
    with torch.no_grad():
        want = cb.decode(codes).numpy()
    got = bias + sum(lut[l][codes[l][:, 0].numpy()] for l in range(3))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)


def test_ptq_state_roundtrip(tmp_path):
    from _common import calibrated_plugin
    from quantv2x_amd.ptq_state import export_ptq_state, load_ptq_state, save_ptq_state
    st = export_ptq_state(calibrated_plugin())
    save_ptq_state(str(tmp_path / "s.npz"), st)
    back = load_ptq_state(str(tmp_path / "s.npz"))
    assert sorted(back) == sorted(st)
    for k in st:
        np.testing.assert_array_equal(back[k], st[k])


def test_export_refuses_networks_the_engine_does_not_build():
    """a QuantModel whose structure the engine hard-wires differently must raise, not deploy as another network."""
    import copy
    import torch
    from _common import build_plugin, quant_wrap, act_quantizers, scene
    from quantv2x_amd import synth
    from quantv2x_amd.plugin.quant import QuantModel
    from quantv2x_amd.plugin.tools import train_utils
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax
    from quantv2x_amd.ptq_state import export_ptq_state

    def plugin(mutate):
        hy = synth.make_hypes("tiny")
        mutate(hy["model"]["args"])
        m = train_utils.create_model(copy.deepcopy(hy)).eval()
        synth.load_state_dict_numpy(m, synth.make_state_dict(m.state_dict(), seed=1))
        return m

    # max fusion (F-Cooper) is built since round 2: it exports, tagged; any other fusion the plugin cannot even construct
    qt = calibrate_minmax(quant_wrap(plugin(lambda a: a.update(fusion_method="max"))), [scene(2)])
    assert str(export_ptq_state(qt)["meta/fusion_method"]) == "max"
    with pytest.raises(NotImplementedError):
        plugin(lambda a: a.update(fusion_method="v2xvit"))
    # post-fusion shrink_conv
    sh = {"kernal_size": [3], "stride": [1], "padding": [1], "dim": [256], "input_dim": 256}
    qt = calibrate_minmax(quant_wrap(plugin(lambda a: a.update(shrink_header=sh))), [scene(2)])
    with pytest.raises(NotImplementedError, match="shrink"):
        export_ptq_state(qt)
    # unfolded BN (is_fusing=False)
    wq = dict(n_bits=8, channel_wise=True, scale_method="minmax")
    aq = dict(n_bits=8, channel_wise=False, scale_method="minmax", leaf_param=True, prob=0.5)
    qt = QuantModel(build_plugin(), wq, aq, is_fusing=False).eval()
    for q in act_quantizers(qt):
        q.set_inited(True)
    for m in qt.modules():
        if hasattr(m, "weight_quantizer"):
            m.weight_quantizer.set_inited(True)
    with pytest.raises(NotImplementedError, match="norm_function"):
        export_ptq_state(qt)
    # disable_act_quant on a backbone layer
    from _common import calibrated_plugin
    qt = calibrated_plugin()
    qt.model.backbone_m1.blocks[0][1].disable_act_quant = True
    with pytest.raises(NotImplementedError, match="disable_act_quant"):
        export_ptq_state(qt)
    # and the golden recipe still exports, carrying the fusion method
    st = export_ptq_state(calibrated_plugin())
    assert str(st["meta/fusion_method"]) == "att"


def test_sharded_driver_refuses_codebookless_engine():
    from quantv2x_amd.dist import AgentShardedModel

    class _E:
        has_codebook = False
    with pytest.raises(NotImplementedError, match="code planes"):
        AgentShardedModel(_E())


def test_adaround_mse_and_output_off_states_export():
    """The export of the three non-default PTQ states equals what the torch quantizers themselves compute, and the oracle
    runs them (the GPU side is tests/test_hip_deploy_states.py)."""
    import torch
    from _common import scene_np
    from _states import adaround_plugin, mse_plugin, output_quant_off_plugin
    from oracle.spec import Oracle
    from quantv2x_amd.plugin.quant import QuantModule
    from quantv2x_amd.ptq_state import export_ptq_state
    qt = adaround_plugin()
    st = export_ptq_state(qt)
    n = 0
    for name, m in qt.model.named_modules():
        if isinstance(m, QuantModule):
            with torch.no_grad():
                want = m.weight_quantizer(m.weight).numpy()                      # floor(w/d) + (alpha >= 0), clamped, dequantized
            shape = [-1] + [1] * (want.ndim - 1)
            got = (st[name + "/w_code"].astype(np.float32) - st[name + "/w_zp"].reshape(shape)) * st[name + "/w_delta"].reshape(shape)
            np.testing.assert_array_equal(got.astype(np.float32), want, err_msg=name)
            nearest = np.clip(np.round(m.weight.detach().numpy() / st[name + "/w_delta"].reshape(shape)) + st[name + "/w_zp"].reshape(shape), 0, 255)
            n += int((nearest != st[name + "/w_code"]).sum())
    assert n > 1000                                   # the hard masks really differ from round-to-nearest
    out = Oracle(st).forward(scene_np(2))
    assert np.isfinite(out["preds_tensor"]).all()
    st = export_ptq_state(output_quant_off_plugin())
    assert bool(st["cls_head/a_off"]) and bool(st["dir_head_single/a_off"]) and not bool(st["backbone_m1.blocks.0.1/a_off"])
    st_mse, st_mm = export_ptq_state(mse_plugin()), export_ptq_state(calibrated_plugin_cached())
    k = "backbone_m1.blocks.1.2/w_delta"
    assert not np.array_equal(st_mse[k], st_mm[k])


def calibrated_plugin_cached():
    from _common import calibrated_plugin
    return calibrated_plugin()


def test_argument_errors_of_the_pyramid_entries_without_gpu():
    """the f3 entry points reject bad descriptors / pointers on the host, before any launch"""
    from quantv2x_amd import lib
    l = _lib()
    buf = np.zeros(4096, np.uint8)
    p = lib.ptr(buf)
    d = lib.Conv1x1Desc()
    d.n, d.h, d.w, d.cin, d.cout, d.stride, d.mode, d.out_ctotal, d.out_delta = 1, 4, 4, 96, 64, 1, 0, 64, 0.1
    assert l.qv2x_conv1x1_i8(C.byref(d), p, p, p, p, p, p, None, p, None) == -2 and b"cin 64" in l.qv2x_last_error()
    d.cin, d.mode = 64, 2
    assert l.qv2x_conv1x1_i8(C.byref(d), p, p, p, p, p, p, None, p, None) == -1 and b"shortcut" in l.qv2x_last_error()
    d.mode, d.stride = 0, 3
    assert l.qv2x_conv1x1_i8(C.byref(d), p, p, p, p, p, p, None, p, None) == -1
    g = lib.GconvDesc()
    g.n, g.h, g.w, g.c, g.cg, g.stride, g.out_delta = 1, 4, 4, 128, 6, 1, 0.1
    assert l.qv2x_gconv3x3_i8(C.byref(g), p, p, p, p, p, p, p, None) == -2 and b"per group" in l.qv2x_last_error()
    g.cg, g.out_delta = 4, 0.0
    assert l.qv2x_gconv3x3_i8(C.byref(g), p, p, p, p, p, p, p, None) == -1 and b"out_delta" in l.qv2x_last_error()
    o = lib.OccDesc()
    o.n, o.h, o.w, o.c, o.out_delta = 1, 4, 4, 60, 0.1
    assert l.qv2x_occ_score_i8(C.byref(o), p, p, p, p, None, None) == -1
    assert l.qv2x_codebook_decode_f32(p, 16, 16, 1, 16, 3, 128, 62, p, p, p, None) == -1            # width % 4
    assert l.qv2x_codebook_decode_f32(p, 16, 16, 1, 16, 17, 128, 64, p, p, p, None) == -1           # planes (levels * seg_num <= 16)
    e = lib.EncodeDesc()
    e.n, e.h, e.w, e.levels, e.kc = 1, 4, 4, 3, 100
    ptrs = (C.c_void_p * 3)()
    assert l.qv2x_codebook_encode64_f32(C.byref(e), 64, p, ptrs, p, None) == -1 and b"dict_size" in l.qv2x_last_error()
    e.kc = 128
    assert l.qv2x_codebook_encode64_f32(C.byref(e), 48, p, ptrs, p, None) == -2                     # fewer than 64 channels
    assert l.qv2x_add_relu_f32(p, p, p, 6, None) == -2                                              # count % 4
    assert l.qv2x_codebook64_level_floats(128) == 3 * (64 * 64 + 64) + 2 * 64 * 128 + 128
    f = lib.FuseDesc()
    f.agents, f.h, f.w, f.max_cav, f.ego, f.h_metres, f.w_metres, f.discrete_ratio = 2, 4, 4, 5, 0, 1.0, 1.0, 1.0
    assert l.qv2x_pyramid_weighted_fuse_i8(C.byref(f), 48, p, 0, 0.1, p, p, p, None) == -1 and b"channels" in l.qv2x_last_error()
    f.fusion = 7
    assert l.qv2x_fuse_att_f32(C.byref(f), None, None, None, p, p, p, None) == -1 and b"fusion" in l.qv2x_last_error()
    pp = lib.PostprocessDesc()
    pp.h, pp.w, pp.anchors_per_cell, pp.num_classes, pp.max_boxes, pp.score_threshold, pp.max_extent, pp.z_min, pp.z_max = 4, 4, 2, 1, 10, 0.2, 6.0, -3.0, 1.0
    assert l.qv2x_postprocess_late_workspace_bytes(C.byref(pp), 9) == -1 and l.qv2x_postprocess_late_workspace_bytes(C.byref(pp), 2) > 0


def test_collapsed_encoder_algebra_matches_the_exact_checker():
    """engine.collapse_encoder (the operands of the OPT-IN qv2x_codebook_encode_collapsed_f32) evaluated in numpy: on the tiny scene every
    index equals the exact checker's, or differs only where the exact top-2 gap is within rounding error."""
    import numpy as np
    from _common import calibrated_plugin, scene_np
    from oracle.spec import Oracle
    from quantv2x_amd.engine import collapse_encoder
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2))
    orc, taps = Oracle(state), {}
    orc.forward(scene_np(2), taps)
    shr, q = taps["shrinker_m1.layers.0.double_conv.1"], taps["shrinker_q"]
    exact, gaps = orc.encode_rows(((shr.astype(np.float32) - np.float32(q[1])) * np.float32(q[0])).reshape(-1, 256), want_gaps=True)
    gp, bias, tab = collapse_encoder(state, 3, float(q[0]), int(q[1]))
    assert gp.shape == (12, 128, 64) and bias.shape == (384,) and tab.shape == (3, 128, 128)
    lane = np.arange(64)
    G = np.zeros((384, 256), np.float32)
    for t in range(12):
        for i in range(128):
            G[t * 32 + (lane & 31), 2 * i + (lane >> 5)] = gp[t, i]
    scores = shr.reshape(-1, 256).astype(np.float32) @ G.T + bias
    codes = np.zeros((3, scores.shape[0]), np.int64)
    for l in range(3):
        v = scores[:, l * 128:(l + 1) * 128].copy()
        for j in range(l):
            v += tab[l * (l - 1) // 2 + j][codes[j]]
        codes[l] = v.argmin(1)
    mm = codes != exact
    assert mm.mean() < 2e-3 and (not mm.any() or gaps[mm].max() < 0.05)
