"""a6, the wave-per-32-cells form of the encode kernel (codebook_encode_wave.hip; whole rounds of the chip's wave slots take it, the remainder
of a launch and launches of less than a round the workgroup form): the SAME codes as the oracle and as the workgroup form -- on the reference's 35 200 golden rows, on ragged launches, on fp32 rows, at every dictionary size."""
import copy
import ctypes as C

import numpy as np
import pytest

from test_codebook_full_golden import _state, golden_rows

pytestmark = pytest.mark.gpu


def _blobs(rng, levels, kc, dev):
    """random level blobs in the library's layout (include/qv2x.h) -- weights of the scale the trained heads have"""
    import torch
    from quantv2x_amd import lib as L
    from quantv2x_amd.engine import _pack_k4p, wave_section
    lib = L.load()
    blobs = []
    for l in range(levels):
        last = l + 1 == levels
        w = [rng.standard_normal((256, 256)).astype(np.float32) / 16 for _ in range(3)]
        b = [rng.standard_normal(256).astype(np.float32) * 0.1 for _ in range(3)]
        if last:
            w[2][:] = 0; b[2][:] = 0
        cb = rng.standard_normal((kc, 256)).astype(np.float32)
        parts = [_pack_k4p(w[0]), b[0], _pack_k4p(w[1]), b[1], _pack_k4p(w[2]), b[2], _pack_k4p(cb), cb, np.zeros(kc, np.float32)]
        flat = np.concatenate([p.reshape(-1) for p in parts])
        wg = flat.size
        flat = np.concatenate([flat, wave_section(w[0], w[1], w[2], cb)])
        assert flat.size == lib.qv2x_codebook_level_floats(kc)
        t = torch.from_numpy(flat).to(dev)
        L.check(lib.qv2x_codebook_c2_f32(C.c_void_p(t.data_ptr() + 4 * (wg - kc - kc * 256)), kc, C.c_void_p(t.data_ptr() + 4 * (wg - kc)), L.current_stream()), "c2")
        blobs.append(t)
    return blobs


@pytest.mark.parametrize("levels,kc,n,h,w", [(3, 128, 1, 7, 13), (3, 128, 2, 9, 33), (2, 96, 1, 8, 40), (1, 32, 3, 5, 7), (3, 64, 1, 33, 31)])
def test_wave_form_equals_workgroup_form_on_random_heads(levels, kc, n, h, w):
    import torch
    from quantv2x_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(levels * 1000 + kc + h)
    blobs = _blobs(rng, levels, kc, dev)
    ptrs = (C.c_void_p * levels)(*[b.data_ptr() for b in blobs])
    x = torch.from_numpy(rng.integers(-128, 128, size=(n, h + 2, w + 2, 256), dtype=np.int8)).to(dev)
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.in_zx, d.in_delta = n, h, w, levels, kc, 117, 0.0173
    M = n * h * w
    got = {}
    for form in ("workgroup", "wave"):
        codes = torch.full((levels, M), 255, dtype=torch.uint8, device=dev)
        if form == "wave":
            L.check(lib.qv2x_codebook_encode_wave_f32(C.byref(d), L.ptr(x), None, ptrs, L.ptr(codes), L.current_stream()), "wave")
        else:
            L.check(lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), ptrs, L.ptr(codes), L.current_stream()), "workgroup")
        got[form] = codes.cpu().numpy()
    assert got["workgroup"].max() < kc
    np.testing.assert_array_equal(got["wave"], got["workgroup"])
    # the same rows as fp32 (the un-quantized model's entry): identical codes again
    xf = ((x.to(torch.float32) + float(128 - 117)) * np.float32(0.0173)).contiguous()
    codes = torch.full((levels, M), 255, dtype=torch.uint8, device=dev)
    L.check(lib.qv2x_codebook_encode_wave_f32(C.byref(d), None, L.ptr(xf), ptrs, L.ptr(codes), L.current_stream()), "wave f32in")
    ref = torch.full((levels, M), 255, dtype=torch.uint8, device=dev)
    L.check(lib.qv2x_codebook_encode_f32in(C.byref(d), L.ptr(xf), ptrs, L.ptr(ref), L.current_stream()), "workgroup f32in")
    np.testing.assert_array_equal(codes.cpu().numpy(), ref.cpu().numpy())


@pytest.mark.parametrize("h,w,what", [(128, 256, "exactly one round of waves, no remainder"),
                                      (100, 328, "one round + a remainder of ONE workgroup"),
                                      (128, 384, "one round + a remainder of 512 workgroups (the most a remainder may have)"),
                                      (116, 424, "one round + 513: the remainder is a second round of waves"),
                                      (97, 331, "ragged: the last workgroup of the remainder is partly past the end")])
def test_split_launch_boundaries(h, w, what):
    """qv2x_codebook_encode_f32 splits a launch into whole rounds of the chip's 1 024 wave slots (wave form) + a remainder of up to 512
    32-cell workgroups: at the boundaries of that rule the codes equal the wave form forced over the whole launch."""
    import torch
    from quantv2x_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the boundaries are those of a 256-CU part")
    rng = np.random.default_rng(h * 1000 + w)
    levels, kc = 2, 64
    blobs = _blobs(rng, levels, kc, dev)
    ptrs = (C.c_void_p * levels)(*[b.data_ptr() for b in blobs])
    x = torch.from_numpy(rng.integers(-128, 128, size=(1, h + 2, w + 2, 256), dtype=np.int8)).to(dev)
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.in_zx, d.in_delta = 1, h, w, levels, kc, 131, 0.021
    M = h * w
    a = torch.full((levels, M), 255, dtype=torch.uint8, device=dev)
    b = torch.full((levels, M), 255, dtype=torch.uint8, device=dev)
    L.check(lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), ptrs, L.ptr(a), L.current_stream()), "split")
    L.check(lib.qv2x_codebook_encode_wave_f32(C.byref(d), L.ptr(x), None, ptrs, L.ptr(b), L.current_stream()), "wave")
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert a.max() < kc, what
    np.testing.assert_array_equal(a, b, err_msg=what)


def test_wave_form_on_the_reference_rows(golden):
    """35 200 real rows x 3 levels: the wave form == the oracle == the workgroup form; against the REFERENCE no index outside its own ties."""
    import torch
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    g = golden["codebook_full"]
    codes_u8, rows = golden_rows(g)
    state = copy.copy(_state())
    state["meta/grid"] = np.array(synth.grid_size(*synth.SHAPES["v2xreal"][:2]), dtype=np.int64)
    state["shrinker_m1.layers.0.double_conv.1/a_delta"] = np.float32(g["in_delta"])
    state["shrinker_m1.layers.0.double_conv.1/a_zp"] = np.float32(g["in_zp"])
    eng = deploy(state=state)
    b = eng._workspace(1)
    b["s1"][0, 1:-1, 1:-1, :] = torch.from_numpy((codes_u8.astype(np.int16) - 128).astype(np.int8).reshape(100, 352, 256)).cuda()
    wg = eng.encode_codes(1).cpu().numpy().reshape(3, -1).copy()
    eng.encode_form = "wave"
    wave = eng.encode_codes(1).cpu().numpy().reshape(3, -1).copy()
    orc = Oracle(state).encode_rows(rows)
    np.testing.assert_array_equal(wave, orc)
    np.testing.assert_array_equal(wave, wg)
    mism = wave != g["codes"]
    assert int((mism & (g["gaps"] > 1e-4)).sum()) == 0


def test_many_frames_launch_takes_the_wave_form_and_is_exact():
    """Six V2X-Real frames in one launch (6 600 waves: six rounds of waves + a remainder of 456 workgroups) == one frame at a time (one round
    of waves + 76 workgroups): the library's split between the two kernel forms is invisible in the codes."""
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    state = copy.copy(_state())
    state["meta/grid"] = np.array(synth.grid_size(*synth.SHAPES["v2xreal"][:2]), dtype=np.int64)
    eng = deploy(state=state)
    n = 6
    b = eng._workspace(n)
    rng = np.random.default_rng(5)
    b["s1"][:, 1:-1, 1:-1, :] = torch.from_numpy(rng.integers(-128, 128, size=(n, 100, 352, 256), dtype=np.int8)).cuda()
    many = eng.encode_codes(n).cpu().numpy().reshape(3, n, -1).copy()
    one = deploy(state=state)
    b1 = one._workspace(1)
    for i in range(n):
        b1["s1"].copy_(b["s1"][i:i + 1])
        np.testing.assert_array_equal(one.encode_codes(1).cpu().numpy().reshape(3, -1), many[:, i])
