"""CPU restatement of the quantized SECOND encoder -- TEST INFRASTRUCTURE, never imported by the product.

PARITY UNPINNED against the reference: ``SECOND.forward`` (opencood/models/heter_encoders.py:66-81) runs on spconv, an un-vendored
pip wheel that is absent from /root/reference and from this image, so no golden vector of the reference's own arithmetic can be
made.  What IS pinned: the sparse-convolution semantics restated here equal ``F.conv3d`` on the densified volume restricted to the
active output sites (tests/test_second_cpu.py), and this integer restatement agrees with the fake-quant torch mirror
(``QuantSECOND`` over ``sub_modules/sparse_ops``) to within one code per layer, teacher-forced.

Arithmetic, per layer ``i`` of ``VoxelBackBone8x`` (sparse_backbone_3d.py:48-91) under ``QuantSpconvModule.forward``
(quant_layer.py:460-490: conv on fake-quantized weights -> BatchNorm1d -> ReLU -> output quantizer):

  layer 0 (fp32 voxel means in):  acc = 0; for k (window offset, z-major) over ACTIVE neighbours, c ascending:
                                  acc = fl(acc + fl(x[c] * wd[k][c][co])),  wd = fl(fl(code - zp_w) * delta_w[co])
  layers 1..11 (uint8 codes in):  T = sum over active neighbours of (x - zp_x) * (w - zp_w)   (exact integer),
                                  acc = fl(fl(T) * fl(delta_w[co] * delta_x))
  then                            y = fl(fl(acc * g[co]) + h[co]);  y = max(y, 0);  code = clamp(rint(y / delta_a) + zp_a, 0, 255)

Active outputs: ``SubMConv3d`` keeps the input's sites; ``SparseConv3d`` activates every output position whose window holds an active
input.  Absent sites are the real value 0, so they add nothing.  ``HeightCompression`` (height_compression.py:12-27) densifies to
``[N, C * D, H, W]`` with channel ``c * D + d``; a cell with no site holds the code ``zp_a`` (real 0).
"""
import numpy as np


def mean_vfe(voxel_features, voxel_num_points):
    """mean_vfe.py:24-30, the sum taken slot by slot in fp32."""
    v = voxel_features.astype(np.float32)
    s = v[:, 0].copy()
    for t in range(1, v.shape[1]):
        s = (s + v[:, t]).astype(np.float32)
    n = np.maximum(voxel_num_points.astype(np.float32), np.float32(1.0)).reshape(-1, 1)
    return (s / n).astype(np.float32)


def _key(b, z, y, x, shape):
    return ((b.astype(np.int64) * shape[0] + z) * shape[1] + y) * shape[2] + x


def out_shape(shape, geom):
    subm, k, s, p = int(geom[0]), geom[1:4], geom[4:7], geom[7:10]
    if subm:
        return [int(v) for v in shape]
    return [int((n + 2 * pp - kk) // ss + 1) for n, pp, kk, ss in zip(shape, p, k, s)]


def out_sites(idx, shape, geom):
    """Active outputs of a strided sparse convolution, raster order of (batch, z, y, x)."""
    k, s, p = geom[1:4], geom[4:7], geom[7:10]
    osh = out_shape(shape, geom)
    keys = []
    i = idx.astype(np.int64)
    for kz in range(k[0]):
        for ky in range(k[1]):
            for kx in range(k[2]):
                num = [i[:, 1 + a] + p[a] - kk for a, kk in enumerate((kz, ky, kx))]
                ok = np.ones(len(i), bool)
                o = []
                for a in range(3):
                    ok &= (num[a] % s[a] == 0)
                    q = num[a] // s[a]
                    ok &= (q >= 0) & (q < osh[a])
                    o.append(q)
                keys.append(_key(i[:, 0], o[0], o[1], o[2], osh)[ok])
    keys = np.unique(np.concatenate(keys))
    x = keys % osh[2]
    y = (keys // osh[2]) % osh[1]
    z = (keys // (osh[2] * osh[1])) % osh[0]
    b = keys // (osh[2] * osh[1] * osh[0])
    return np.stack([b, z, y, x], 1).astype(np.int32)


def rulebook(oidx, iidx, ishape, geom):
    """nbr [K, N_out]: row of the input site under window offset k of each output, -1 when absent."""
    k, s, p = geom[1:4], geom[4:7], geom[7:10]
    ikeys = _key(iidx[:, 0], iidx[:, 1].astype(np.int64), iidx[:, 2].astype(np.int64), iidx[:, 3].astype(np.int64), ishape)
    order = np.argsort(ikeys)
    skeys = ikeys[order]
    o = oidx.astype(np.int64)
    nbr = np.full((int(k[0] * k[1] * k[2]), len(o)), -1, np.int64)
    kk = 0
    for kz in range(k[0]):
        for ky in range(k[1]):
            for kx in range(k[2]):
                src = [o[:, 1 + a] * s[a] - p[a] + t for a, t in enumerate((kz, ky, kx))]
                ok = np.ones(len(o), bool)
                for a in range(3):
                    ok &= (src[a] >= 0) & (src[a] < ishape[a])
                q = _key(o[:, 0], src[0], src[1], src[2], ishape)
                if len(skeys):
                    pos = np.minimum(np.searchsorted(skeys, q), len(skeys) - 1)
                    hit = ok & (skeys[pos] == q)
                    nbr[kk, hit] = order[pos[hit]]
                kk += 1
    return nbr


def _finish(acc, st, p):
    y = (acc * st[p + "bn_g"][None, :]).astype(np.float32)
    y = (y + st[p + "bn_h"][None, :]).astype(np.float32)
    y = np.maximum(y, np.float32(0.0))
    d = np.float32(st[p + "a_delta"])
    return np.clip(np.rint((y / d).astype(np.float32)) + np.float32(st[p + "a_zp"]), 0, 255).astype(np.uint8)


class OracleSecond:
    def __init__(self, state):
        self.st = state
        self.n_layers = int(state["second/n_layers"])
        self.shape0 = [int(v) for v in state["second/sparse_shape"]]

    def layer(self, i, x, idx, shape, forced_in=None):
        """One sparse layer: ``x`` = fp32 means (i == 0) or uint8 codes; returns (codes u8 [N_out, C_out], out idx, out shape)."""
        st, p = self.st, f"second/{i}/"
        geom = [int(v) for v in st[p + "geom"]]
        oidx = idx if geom[0] else out_sites(idx, shape, geom)
        nbr = rulebook(oidx, idx, shape, geom)
        wc = st[p + "w_code"].astype(np.int64)                        # [K, Cin, Cout]
        zw = st[p + "w_zp"].astype(np.int64)
        co = wc.shape[2]
        if i == 0:
            wd = ((wc - zw[None, None, :]).astype(np.float32) * st[p + "w_delta"][None, None, :]).astype(np.float32)
            acc = np.zeros((len(oidx), co), np.float32)
            for k in range(wc.shape[0]):
                hit = nbr[k] >= 0
                rows = nbr[k][hit]
                for c in range(wc.shape[1]):
                    prod = (x[rows, c][:, None] * wd[k, c][None, :]).astype(np.float32)
                    acc[hit] = (acc[hit] + prod).astype(np.float32)
        else:
            zx = int(st[f"second/{i - 1}/a_zp"])
            T = np.zeros((len(oidx), co), np.float64)                 # exact: |T| < 2^31
            wz = (wc - zw[None, None, :]).astype(np.float64)
            xz = x.astype(np.float64) - zx
            for k in range(wc.shape[0]):
                hit = nbr[k] >= 0
                if hit.any():
                    T[hit] += xz[nbr[k][hit]] @ wz[k]
            sc = (st[p + "w_delta"].astype(np.float32) * np.float32(st[f"second/{i - 1}/a_delta"])).astype(np.float32)
            acc = (T.astype(np.float32) * sc[None, :]).astype(np.float32)
        return _finish(acc, st, p), oidx, out_shape(shape, geom)

    def forward(self, inputs, taps=None, batch_size=None):
        """``inputs``: voxel_features / voxel_coords (agent, z, y, x) / voxel_num_points -> dense uint8 BEV codes ``[N, C * D, H, W]``."""
        idx = inputs["voxel_coords"].astype(np.int32)
        n = int(idx[:, 0].max()) + 1 if batch_size is None else int(batch_size)
        x = mean_vfe(inputs["voxel_features"], inputs["voxel_num_points"])
        if taps is not None:
            taps["mean_vfe"] = x
        shape = self.shape0
        for i in range(self.n_layers):
            x, idx, shape = self.layer(i, x, idx, shape)
            if taps is not None:
                taps[f"second/{i}"] = (x, idx, list(shape))
        last = f"second/{self.n_layers - 1}/"
        c, (d, h, w) = x.shape[1], shape
        vol = np.full((n, c, d, h, w), np.uint8(int(self.st[last + "a_zp"])), np.uint8)
        vol[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = x
        return vol.reshape(n, c * d, h, w)

    def dequant(self, bev):
        last = f"second/{self.n_layers - 1}/"
        return ((bev.astype(np.float32) - np.float32(self.st[last + "a_zp"])) * np.float32(self.st[last + "a_delta"])).astype(np.float32)
