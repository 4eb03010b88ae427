"""qv2x_deconv_i8_batch (round 5: pixel-stationary items, csrc/deconv_f32.hip) against oracle/qv2x_oracle.c:orc_deconv and against the
one-layer entry qv2x_deconv_i8, directly on random layers: every Cin the pixel-stationary form takes (64 | 128 | 256) with s = 1 | 2 | 4,
pixel counts that are no multiple of a tile, every pairs-per-item choice of the launch rule (the batch size decides it), a layer it does
NOT take (Cin 32: the batch falls back to the 32 x 32 wave tiles), destinations with a channel window.  Bit-exact.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(rng, cin, cout, s):
    from quantv2x_amd.engine import _pack_k4p
    wdeq = (rng.standard_normal((cin, cout, s, s)) * (0.6 / np.sqrt(cin))).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    cols = wdeq.transpose(0, 2, 3, 1).reshape(cin, s * s * cout)              # col = (i*s + j)*Cout + co
    return wdeq, bias, _pack_k4p(np.ascontiguousarray(cols.T))


def _padded(x_u8, pad_code):
    n, h, w, c = x_u8.shape
    out = np.full((n, h + 2, w + 2, c), pad_code, np.uint8)
    out[:, 1:-1, 1:-1] = x_u8
    return (out.astype(np.int16) - 128).astype(np.int8)


def _run(layers, n, relu=1, use_batch=True):
    """layers: [(cin, cout, s, h, w)] writing side by side into one [n][H+2][W+2][sum cout + 16] destination (16 spare channels in front)"""
    from oracle.spec import _cf, _f32, _p, lib as olib
    from quantv2x_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(sum(a * 7 + b for (a, b, *_r) in layers) + n)
    H, W = layers[0][3] * layers[0][2], layers[0][4] * layers[0][2]
    ctot = 16 + sum(l[1] for l in layers)
    want = np.zeros((n, H, W, ctot), np.uint8)
    out = torch.full((n, H + 2, W + 2, ctot), -128, dtype=torch.int8, device=dev)
    descs, ins, ws, bs, keep = [], [], [], [], []
    c0 = 16
    for (cin, cout, s, h, w) in layers:
        assert (h * s, w * s) == (H, W)
        wdeq, bias, packed = _layer(rng, cin, cout, s)
        x = rng.integers(0, 256, size=(n, h, w, cin), dtype=np.uint8)
        zx, dx, da, za = int(rng.integers(0, 200)), 0.031, 0.047, float(rng.integers(0, 3))
        olib().orc_deconv(_p(x), n, h, w, cin, _cf(dx), zx, _p(wdeq), _p(_f32(bias)), cout, s, relu, _cf(da), _cf(za), _p(want), ctot, c0)
        d = L.DeconvDesc()
        d.n, d.h, d.w, d.cin, d.cout, d.s, d.in_zx, d.in_delta = n, h, w, cin, cout, s, zx, dx
        d.out_ctotal, d.out_c0, d.relu, d.out_delta, d.out_zp, d.out_h, d.out_w = ctot, c0, relu, da, za, H, W
        xt = torch.from_numpy(_padded(x, zx)).to(dev)
        wt, bt = torch.from_numpy(packed).to(dev), torch.from_numpy(bias).to(dev)
        keep += [xt, wt, bt]
        descs.append(d); ins.append(xt); ws.append(wt); bs.append(bt)
        c0 += cout
    k = len(layers)
    if use_batch:
        arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])
        L.check(lib.qv2x_deconv_i8_batch((L.DeconvDesc * k)(*descs), k, arr(ins), arr(ws), arr(bs), arr([out] * k), L.current_stream()), "batch")
    else:
        for d, xt, wt, bt in zip(descs, ins, ws, bs):
            L.check(lib.qv2x_deconv_i8(C.byref(d), L.ptr(xt), L.ptr(wt), L.ptr(bt), L.ptr(out), L.current_stream()), "single")
    torch.cuda.synchronize()
    got = (out[:, 1:-1, 1:-1].cpu().numpy().astype(np.int16) + 128).astype(np.uint8)
    return got, want


# (n frames decide the pairs-per-item rule: a few items -> 1 or 2 pairs, many -> 8)
@pytest.mark.parametrize("n,layers", [
    (1, [(64, 128, 1, 20, 44), (128, 128, 2, 10, 22), (256, 128, 4, 5, 11)]),        # the backbone's three deblocks, small maps, ragged tiles
    (3, [(64, 128, 1, 36, 52), (128, 128, 2, 18, 26), (256, 128, 4, 9, 13)]),
    (2, [(256, 64, 2, 33, 47)]),                                                      # one layer through the batch entry, 64-column chunks
    (2, [(128, 64, 1, 31, 33), (64, 192, 1, 31, 33)]),                                # 64 / 192 columns: one pair, and three (items of ONE pair each)
    (2, [(32, 64, 2, 15, 17), (64, 64, 2, 15, 17)]),                                  # Cin 32: not a pixel-stationary shape -> the wave-tile form
    (8, [(32, 64, 2, 64, 64), (96, 64, 2, 64, 64)]),                                  # ... and its 64 x 64-tile variant (>= 16384 wave tiles)
])
def test_batch_equals_oracle_and_single(n, layers):
    got, want = _run(layers, n)
    np.testing.assert_array_equal(got[..., 16:], want[..., 16:])
    assert (got[..., :16] == 0).all()                                                 # the channel window in front stays untouched (-128 = code 0)
    one, _ = _run(layers, n, use_batch=False)
    np.testing.assert_array_equal(one, got)


def test_many_items_take_eight_pairs_per_item():
    """enough pixels for the rule's longest items (finding 14): 12 frames of 50 x 88 -> 8 column pairs per item at Cin 128 | 256"""
    got, want = _run([(64, 128, 1, 100, 88), (128, 128, 2, 50, 44), (256, 128, 4, 25, 22)], 12, relu=0)
    np.testing.assert_array_equal(got[..., 16:], want[..., 16:])


@pytest.mark.parametrize("cin,cout,s,n,h,w", [(64, 128, 1, 2, 37, 45), (128, 128, 2, 2, 19, 23), (256, 128, 4, 1, 9, 11), (64, 64, 1, 1, 5, 7), (32, 64, 2, 1, 9, 9)])
def test_f32in_equals_oracle(cin, cout, s, n, h, w):
    """qv2x_deconv_f32in (the Pyramid model's deblocks and first 1x1 on fp32 maps): the pixel-stationary items for 64 / 128 input channels
    (round 5), the wave-tile kernel for the rest -- against orc_deconv_f32in, bit for bit"""
    from oracle.spec import _cf, _f32, _p, lib as olib
    from quantv2x_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(cin + cout + s + n)
    wdeq, bias, packed = _layer(rng, cin, cout, s)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    H, W, ctot, c0, da, za = h * s, w * s, cout + 16, 16, 0.043, 1.0
    want = np.zeros((n, H, W, ctot), np.uint8)
    olib().orc_deconv_f32in(_p(x), n, h, w, cin, _p(wdeq), _p(_f32(bias)), cout, s, 1, _cf(da), _cf(za), _p(want), ctot, c0)
    out = torch.full((n, H + 2, W + 2, ctot), -128, dtype=torch.int8, device=dev)
    d = L.DeconvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.s, d.in_zx, d.in_delta = n, h, w, cin, cout, s, 0, 1.0
    d.out_ctotal, d.out_c0, d.relu, d.out_delta, d.out_zp, d.out_h, d.out_w = ctot, c0, 1, da, za, H, W
    xt, wt, bt = torch.from_numpy(x).to(dev), torch.from_numpy(packed).to(dev), torch.from_numpy(bias).to(dev)
    L.check(lib.qv2x_deconv_f32in(C.byref(d), L.ptr(xt), L.ptr(wt), L.ptr(bt), L.ptr(out), L.current_stream()), "f32in")
    torch.cuda.synchronize()
    got = (out[:, 1:-1, 1:-1].cpu().numpy().astype(np.int16) + 128).astype(np.uint8)
    np.testing.assert_array_equal(got[..., 16:], want[..., 16:])
    assert (got[..., :16] == 0).all()
