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
    assert l.qv2x_version() == 1


def test_struct_layouts_match_header():
    from quantv2x_amd import lib
    assert C.sizeof(lib.PfnParams) == (640 + 64 + 4 + 6) * 4
    assert C.sizeof(lib.ConvDesc) == (7 + 3 * 4 + 3) * 4 + 8
    assert C.sizeof(lib.DeconvDesc) == 13 * 4
    assert C.sizeof(lib.EncodeDesc) == 7 * 4
    assert C.sizeof(lib.FuseDesc) == 7 * 4 + 4 + 2 * 8 + 3 * 8      # 4 bytes of padding before the int64 fields


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
    assert l.qv2x_codebook_level_floats(128) == 3 * (65536 + 256) + 256 * 128 * 2 + 128


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
