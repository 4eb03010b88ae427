"""qv2x_conv3x3_i8_wide (halo-tiled, pre-tiled weights) vs qv2x_conv3x3_i8 and the CPU oracle: bit-exact uint8 codes.

Covers the shapes the end-to-end tests do not reach at small size: one exact 5 x 32 patch, ragged patches on both
axes, several images, one and three input groups (the concat of the three deblocks, three activation scales)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer_state(rng, name, cin, cout):
    return {
        name + "/w_code": rng.integers(0, 256, size=(cout, cin, 3, 3), dtype=np.uint8),
        name + "/w_delta": rng.uniform(0.002, 0.01, size=cout).astype(np.float32),
        name + "/w_zp": rng.integers(100, 156, size=cout).astype(np.float32),
        name + "/bias": rng.normal(0, 0.5, size=cout).astype(np.float32),
        name + "/a_delta": np.float32(0.11), name + "/a_zp": np.float32(3.0), name + "/a_off": np.bool_(False),
    }


def _padded(x_u8, zx_per_channel):
    """u8 [N,H,W,C] -> padded i8 [N,H+2,W+2,C] with border = zero point (x - zx == 0 there)"""
    n, h, w, c = x_u8.shape
    p = np.empty((n, h + 2, w + 2, c), np.int16)
    p[:] = zx_per_channel.astype(np.int16)
    p[:, 1:-1, 1:-1] = x_u8
    return (p - 128).astype(np.int8)


@pytest.mark.parametrize("n,h,w,groups,cout", [
    (1, 5, 32, [(0, 64)], 256),
    (1, 10, 16, [(0, 64)], 256),
    (1, 20, 32, [(0, 128)], 256),
    (2, 13, 37, [(0, 64), (64, 128), (192, 64)], 256),
    (1, 7, 50, [(0, 128), (128, 64)], 256),
    (3, 25, 16, [(0, 256)], 256),
    (40, 25, 32, [(0, 256)], 256),          # 200 patches: the eight-wave, 256-channel workgroups
    (1, 10, 16, [(0, 64)], 64),             # the backbone's 64- and 128-channel layers (two / four waves per workgroup)
    (2, 13, 37, [(0, 64)], 64),
    (1, 5, 32, [(0, 128)], 128),
    (2, 13, 37, [(0, 128)], 128),
    (1, 6, 33, [(0, 192)], 256),            # an odd number of 64-channel chunks (the halo double buffer ends on buffer 0)
    (1, 4, 64, [(0, 64), (64, 64)], 256),   # one-chunk groups: a group fold after every chunk
    (1, 7, 40, [(0, 64), (64, 64), (128, 64), (192, 64)], 256),   # QV2X_MAX_GROUPS groups: every slot of the window-sum table
    (5, 9, 70, [(0, 128)], 64),             # two-wave workgroups looping over the halo pixels, two chunks
    (1, 5, 32, [(0, 1536)], 256),           # MAX_CHUNKS chunks
    (12, 50, 96, [(0, 64)], 768),           # 3 channel blocks x 360 patches: persistent workgroups take several items each; the grid
                                            # must stay a multiple of 8 x 3 for a workgroup to keep its channel block
    (30, 50, 64, [(0, 64)], 64),            # 600 one-chunk patches on 768 two-wave slots ... and
    (80, 50, 64, [(0, 64)], 64),            # 1600: two or three items per workgroup, the next-next tile requested at the rotation
    (140, 25, 64, [(0, 64)], 64),           # 1400 patches on 1024 ping-pong groups: one or two items per group, idle tail slots
    (64, 50, 96, [(0, 128)], 128),          # 1920 items on 512 four-wave groups: 3 or 4 each, two chunks (three slots per item)
    (24, 25, 88, [(0, 256)], 256),          # the 25 x 88 level: 360 patches x two 128-channel blocks on 512 groups, four chunks
    (2, 10, 40, [(0, 192)], 128),           # three chunks (odd): the late half starts two slots behind; most groups idle
    # round 4: the weights-stationary form (conv_i8_ws.hip: 64 -> 64 channels, stride 1, from 1024 patches of 10 x 32 on)
    (90, 33, 70, [(0, 64)], 64),            # 1080 patches, ragged on both axes (3 of 10 rows, 6 of 32 columns): 2 or 3 items per workgroup
    (10, 100, 352, [(0, 64)], 64),          # the V2X-Real level-0 map: 10 x 11 exact patches per frame
    (1100, 7, 20, [(0, 64)], 64),           # one mostly empty patch per image: the second row group stores two rows of five
    (32, 101, 65, [(0, 64)], 64),           # 11 x 3 patches per image, one row and one column into the last patch
])
def test_wide_matches_regular_and_oracle(n, h, w, groups, cout):
    from oracle.spec import Oracle
    from quantv2x_amd import lib as L
    from quantv2x_amd.engine import _ConvLayer
    lib = L.load()
    dev = torch.device("cuda")
    rng = np.random.default_rng(h * 1000 + w)
    cin, name = sum(c for _, c in groups), "layer"
    state = _layer_state(rng, name, cin, cout)
    in_q = [(c0, c, np.float32(0.03 + 0.02 * i), 90 + 30 * i) for i, (c0, c) in enumerate(groups)]
    layer = _ConvLayer(state, name, in_q, 1, dev)
    x = rng.integers(0, 256, size=(n, h, w, cin), dtype=np.uint8)
    zx = np.concatenate([np.full(c, z) for (_, c, _, z) in in_q])
    xin = torch.from_numpy(_padded(x, zx)).to(dev)

    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin_total, d.stride, d.cout = n, h, w, cin, 1, cout
    d.ngroups = len(groups)
    for i, (c0, c, _, z) in enumerate(in_q):
        d.group_c0[i], d.group_c[i], d.group_zx[i] = c0, c, z
    d.out_ctotal, d.out_c0, d.relu = cout, 0, 1
    d.out_delta, d.out_zp = layer.out_q[0], float(layer.out_q[1])
    st = L.current_stream()
    outs = []
    # the regular (im2col) kernel, then the halo-patch one
    forms = [None, 0]
    for form in forms:
        out = torch.full((n, h + 2, w + 2, cout), -77, dtype=torch.int8, device=dev)
        if form is not None:
            w_wide = torch.empty_like(layer.w)
            L.check(lib.qv2x_conv3x3_i8_pack_wide(C.byref(d), L.ptr(layer.w), L.ptr(w_wide), st), "pack")
            L.check(lib.qv2x_conv3x3_i8_wide(C.byref(d), L.ptr(xin), L.ptr(w_wide), L.ptr(layer.scale), L.ptr(layer.corr),
                                             L.ptr(layer.aw), L.ptr(layer.bias), L.ptr(out), st), "wide")
        else:
            L.check(lib.qv2x_conv3x3_i8(C.byref(d), L.ptr(xin), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr),
                                        L.ptr(layer.aw), L.ptr(layer.bias), L.ptr(out), st), "regular")
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    for form, o in zip(forms[1:], outs[1:]):
        np.testing.assert_array_equal(o, outs[0], err_msg=f"form {form}")   # interior AND the untouched border
    assert (outs[1][:, 0] == -77).all() and (outs[1][:, :, 0] == -77).all()

    orc = Oracle.__new__(Oracle)
    orc.s = state
    want, _ = orc.conv(name, x, in_q)
    got = (outs[1][:, 1:-1, 1:-1].astype(np.int16) + 128).astype(np.uint8)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n,h,w,cin,cout", [
    (1, 10, 64, 64, 64),            # one exact patch row (5 x 32 outputs), one period of plane chunks
    (2, 27, 75, 64, 64),            # ragged output map (14 x 38)
    (3, 20, 130, 64, 128),          # level 1's first layer: 64 -> 128
    (2, 50, 66, 128, 256),          # level 2's: two periods, eight plane chunks
    (40, 50, 64, 64, 64),           # 1000 patches on 768 slots: the next item's first tile requested from a one-step chunk
    (16, 50, 176, 128, 256),        # 480 patches: eight-wave workgroups, persistent items
    (52, 100, 190, 64, 64),         # round 5: 1092 items of 8 x 32 outputs -> the weights-stationary stride-2 form (conv_i8_ws2.hip), ragged both ways
    (52, 100, 190, 64, 128),        # the same form with four channel blocks (level 1's 64 -> 128)
    (120, 34, 130, 64, 64),         # 1080 items, 17 x 65 outputs: a last tile row of ONE output row, a last tile column of one pixel
    (40, 64, 254, 64, 128),         # 32 x 127 outputs: exact tile rows, the extra column (input column 64 of a tile) inside the map's last tile
])
def test_wide_stride2_matches_regular_and_oracle(n, h, w, cin, cout):
    """The ZeroPad2d + stride-2 first convolution of a backbone level on the halo-patch kernel (parity-plane chunks): equal to the im2col
    kernel and to the oracle, bit for bit."""
    from oracle.spec import Oracle
    from quantv2x_amd import lib as L
    from quantv2x_amd.engine import _ConvLayer
    lib = L.load()
    dev = torch.device("cuda")
    rng = np.random.default_rng(h * 1000 + w + cin)
    name = "layer"
    state = _layer_state(rng, name, cin, cout)
    in_q = [(0, cin, np.float32(0.04), 77)]
    layer = _ConvLayer(state, name, in_q, 2, dev)
    x = rng.integers(0, 256, size=(n, h, w, cin), dtype=np.uint8)
    xin = torch.from_numpy(_padded(x, np.full(cin, 77))).to(dev)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin_total, d.stride, d.cout, d.ngroups = n, h, w, cin, 2, cout, 1
    d.group_c0[0], d.group_c[0], d.group_zx[0] = 0, cin, 77
    d.out_ctotal, d.out_c0, d.relu = cout, 0, 1
    d.out_delta, d.out_zp = layer.out_q[0], float(layer.out_q[1])
    st = L.current_stream()
    outs = []
    for wide in (False, True):
        out = torch.full((n, ho + 2, wo + 2, cout), -77, dtype=torch.int8, device=dev)
        if wide:
            w_wide = torch.empty_like(layer.w)
            L.check(lib.qv2x_conv3x3_i8_pack_wide(C.byref(d), L.ptr(layer.w), L.ptr(w_wide), st), "pack")
            L.check(lib.qv2x_conv3x3_i8_wide(C.byref(d), L.ptr(xin), L.ptr(w_wide), L.ptr(layer.scale), L.ptr(layer.corr),
                                             L.ptr(layer.aw), L.ptr(layer.bias), L.ptr(out), st), "wide")
        else:
            L.check(lib.qv2x_conv3x3_i8(C.byref(d), L.ptr(xin), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr),
                                        L.ptr(layer.aw), L.ptr(layer.bias), L.ptr(out), st), "regular")
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    np.testing.assert_array_equal(outs[1], outs[0])
    orc = Oracle.__new__(Oracle)
    orc.s = state
    want, _ = orc.conv(name, x, in_q, stride=2)
    np.testing.assert_array_equal((outs[1][:, 1:-1, 1:-1].astype(np.int16) + 128).astype(np.uint8), want)


def test_wide_rejects_unsupported():
    from quantv2x_amd import lib as L
    lib = L.load()
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin_total, d.stride, d.cout, d.ngroups = 1, 100, 352, 64, 2, 256, 1
    d.group_c[0], d.out_ctotal, d.out_delta = 64, 256, 0.1
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 0               # stride 2 on one frame: 55 patches do not fill the chip
    d.stride = 3
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 0
    d.stride, d.cout = 1, 192
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 0               # cout: 64, 128 or a multiple of 256
    d.cout = 128
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 1               # 220 patches of 128 channels
    d.cout = 64
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 0               # 220 two-wave workgroups do not fill the chip ...
    d.n = 8
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 1               # ... 1760 do
    d.n, d.cout = 1, 256
    assert lib.qv2x_conv3x3_i8_wide_ok(C.byref(d)) == 1
    x = torch.zeros(64, dtype=torch.int8, device="cuda")
    d.stride = 3
    rc = lib.qv2x_conv3x3_i8_wide(C.byref(d), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), None)
    assert rc != 0 and b"stride" in lib.qv2x_last_error()
    d.stride, d.ngroups, d.cin_total, d.cout = 2, 2, 128, 256
    d.group_c0[1], d.group_c[1] = 64, 64
    rc = lib.qv2x_conv3x3_i8_wide(C.byref(d), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), None)
    assert rc != 0 and b"one input group" in lib.qv2x_last_error()      # the stride-2 form folds one group only
