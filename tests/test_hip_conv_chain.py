"""qv2x_conv3x3_i8_chain64 (several 64-channel conv layers of a backbone level in one launch, intermediates in LDS) vs
the same layers as separate qv2x_conv3x3_i8 launches and vs the CPU oracle: bit-exact uint8 codes.

Shapes the end-to-end frames do not reach: every depth 1..4, both first-layer strides, ragged patches on both axes (the
workgroup patch is 5 x 32), maps smaller than one patch, odd input sizes under stride 2, several images, non-zero
output zero points (the pad code written for out-of-image positions of an intermediate map)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_hip_conv_wide import _layer_state, _padded

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,h_in,w_in,depth,stride0", [
    (1, 10, 64, 4, 2),       # exactly one 5 x 32 patch after the stride-2 layer
    (1, 9, 61, 4, 2),        # odd input: (9 + 2 - 3) / 2 + 1 = 5 rows, 31 columns
    (2, 23, 70, 3, 2),       # ragged on both axes, two images
    (1, 5, 32, 2, 1),
    (1, 13, 37, 3, 1),
    (1, 3, 4, 4, 1),         # map smaller than the halo
    (3, 12, 40, 1, 2),
    (1, 17, 33, 1, 1),
    (1, 40, 100, 4, 2),
])
def test_chain_matches_layerwise_and_oracle(n, h_in, w_in, depth, stride0):
    from oracle.spec import Oracle
    from quantv2x_amd import lib as L
    from quantv2x_amd.engine import _ChainLayers, _ConvLayer
    lib = L.load()
    dev = torch.device("cuda")
    rng = np.random.default_rng(h_in * 1000 + w_in * 10 + depth)
    state, layers = {}, []
    q = (np.float32(0.04), 7)                                           # quantizer of the chain input
    in_q0 = q
    for l in range(depth):
        name = f"l{l}"
        st = _layer_state(rng, name, 64, 64)
        st[name + "/w_delta"] = rng.uniform(0.0005, 0.002, size=64).astype(np.float32)
        st[name + "/a_delta"] = np.float32(0.02 + 0.01 * l)
        st[name + "/a_zp"] = np.float32([3.0, 0.0, 11.0, 5.0][l])      # a non-zero pad code for the next layer
        state.update(st)
        layer = _ConvLayer(state, name, [(0, 64, q[0], q[1])], stride0 if l == 0 else 1, dev)
        layers.append(layer)
        q = layer.out_q
    assert _ChainLayers.eligible(layers)
    chain = _ChainLayers(layers, dev)
    x = rng.integers(0, 256, size=(n, h_in, w_in, 64), dtype=np.uint8)
    xin = torch.from_numpy(_padded(x, np.full(64, in_q0[1]))).to(dev)
    h, w = (h_in + 2 - 3) // stride0 + 1, (w_in + 2 - 3) // stride0 + 1
    st = L.current_stream()

    # layer by layer: padded tensors whose borders hold each layer's own pad code
    cur, hh, ww = xin, h_in, w_in
    for l, layer in enumerate(layers):
        out = torch.full((n, h + 2, w + 2, 64), int(layer.out_q[1]) - 128, dtype=torch.int8, device=dev)
        d = L.ConvDesc()
        d.n, d.h, d.w, d.cin_total, d.stride, d.cout, d.ngroups = n, hh, ww, 64, layer.stride, 64, 1
        d.group_c0[0], d.group_c[0], d.group_zx[0] = 0, 64, layer.groups[0][2]
        d.out_ctotal, d.out_c0, d.relu = 64, 0, 1
        d.out_delta, d.out_zp = layer.out_q[0], float(layer.out_q[1])
        L.check(lib.qv2x_conv3x3_i8(C.byref(d), L.ptr(cur), L.ptr(layer.w), L.ptr(layer.scale), L.ptr(layer.corr),
                                    L.ptr(layer.aw), L.ptr(layer.bias), L.ptr(out), st), layer.name)
        cur, hh, ww = out, h, w
    torch.cuda.synchronize()
    want = cur.cpu().numpy()

    got_t = torch.full((n, h + 2, w + 2, 64), -77, dtype=torch.int8, device=dev)
    cd = L.ChainDesc()
    cd.n, cd.h, cd.w, cd.in_h, cd.in_w, cd.depth, cd.stride0, cd.relu = n, h, w, h_in, w_in, depth, stride0, 1
    for l, layer in enumerate(layers):
        cd.out_delta[l], cd.out_zp[l] = layer.out_q[0], float(layer.out_q[1])
    L.check(lib.qv2x_conv3x3_i8_chain64(C.byref(cd), L.ptr(xin), L.ptr(chain.w), L.ptr(chain.scale), L.ptr(chain.corr), L.ptr(chain.aw),
                                        L.ptr(chain.bias), L.ptr(got_t), st), "chain")
    torch.cuda.synchronize()
    got = got_t.cpu().numpy()
    np.testing.assert_array_equal(got[:, 1:-1, 1:-1], want[:, 1:-1, 1:-1])
    assert (got[:, 0] == -77).all() and (got[:, -1] == -77).all() and (got[:, :, 0] == -77).all() and (got[:, :, -1] == -77).all()

    # and the oracle, layer by layer from the same input
    orc = Oracle.__new__(Oracle)
    orc.s = state
    y, yq = x, in_q0
    for l in range(depth):
        y, yq = orc.conv(f"l{l}", y, [(0, 64, yq[0], yq[1])], stride=stride0 if l == 0 else 1)
    np.testing.assert_array_equal((got[:, 1:-1, 1:-1].astype(np.int16) + 128).astype(np.uint8), y)


def test_chain_rejects_unsupported():
    from quantv2x_amd import lib as L
    lib = L.load()
    x = torch.zeros(64, dtype=torch.int8, device="cuda")
    d = L.ChainDesc()
    d.n, d.h, d.w, d.in_h, d.in_w, d.depth, d.stride0 = 1, 5, 32, 10, 64, 5, 2
    for l in range(4):
        d.out_delta[l] = 0.1
    args = [L.ptr(x)] * 7
    assert lib.qv2x_conv3x3_i8_chain64(C.byref(d), *args, None) == -1 and b"layers" in lib.qv2x_last_error()
    d.depth, d.in_h = 4, 11                       # 11 rows at stride 2 give 6, not 5
    assert lib.qv2x_conv3x3_i8_chain64(C.byref(d), *args, None) == -1 and b"does not give" in lib.qv2x_last_error()
    d.in_h, d.stride0 = 10, 3
    assert lib.qv2x_conv3x3_i8_chain64(C.byref(d), *args, None) == -1
