"""Two-stage EXACT codebook encode (a6), host side: the operands of the candidate stage and the proof obligation that makes its answers final.

``UMGMQuantizer.encode`` (opencood/models/sub_modules/codebook.py:330-337 -> :231-239 -> :106-131) is, per level,
``z = stage(x); q = qhead(z); code = argmin_k (|q|^2 + |c_k|^2) - 2 q . c_k; x <- lhead(z) - c_code`` -- every head affine.  The shipped
exact kernels evaluate that chain in the reference's op order in fp32 (eleven 256-wide GEMMs per cell: 21.9 GMAC per agent-frame, 72 % of a
step).  Stage 1 here evaluates the SAME argmin in exact integer arithmetic on the collapsed form

    dist_l[k] - |q_l|^2  =  s_l[k]  =  G_l[k] . x_0 + g_l[k] + sum_{j<l} T_lj[code_j][k]               (``engine.collapse_encoder``)

(one 256 -> levels * kc product per cell, x_0 = delta * (code - zx) with integer codes, G on a fixed-point grid of 24 bits split into three
int8 limbs: ``v_mfma_i32_32x32x32_i8``, no rounding anywhere) and keeps, per level, the gap between its best and second-best score.  Stage 2
re-runs ONLY the cells whose gap at any level is not larger than a bound tau_l that PROVABLY covers every way the fp32 chain can differ
from real arithmetic -- through the bit-exact kernel (``codebook_encode_wave_kernel`` in list mode).  A cell that passes has, at every
level, ``D_hat[a] < D_hat[b]`` for its candidate a and every b != a in the fp32 chain itself, so the reference-order kernel would have
produced the same index whatever its tie rule: the encode is exact BY CONSTRUCTION, not by measurement.

The bound (all norms 2-norms, u = 2^-24, every fp32 dot product of the chain an ascending-k fma chain with acc0 = bias: what the oracle and
the MFMA kernels evaluate):

* one layer ``v_out = fl(W v_in + b)``: the computed partial sums obey ``s_i = (s_{i-1} + w_i v_i)(1 + t_i)``, ``|t_i| <= u``, so the
  layer's rounding error vector delta has ``||delta|| <= u' sum_i ||s_i|| <= u'' (256 ||b|| + kappa(W) ||v_in||)`` with
  ``kappa(W) = sum_i sigma_max(W[:, :i])`` (the partial sums of ALL outputs as a vector: ``||b + W[:, :i] v[:i]||``), u'' = u inflated for
  the accumulated errors inside the partial sums (1 / (1 - 256 u')).
* the layers are affine and both chains use the same codes, so the difference between the computed q_l and the ideal one is EXACTLY
  ``sum_t P_t delta_t`` (P_t = the product of the weight matrices between source t and q_l) -- no first-order truncation -- and its effect on
  the comparison of codes a and b is ``2 (c_a - c_b) . P_t delta_t <= 2 ||P_t^T (c_a - c_b)|| ||delta_t||``: the sensitivity of FIXED
  directions (maximised over the pairs), not an operator norm.  Sources: the input's own rounding ``fl(n delta)``, stage_j / lhead_j (+ the
  residual subtraction) of every level in front, qhead_l.
* the distance itself: ``2 * 256 u'' ||c_k|| ||q||`` for the chain of q . c_k, and ``u (2 + u)(X2 + |c_k|^2) + 2 u |I_k|`` for the two roundings of
  ``fl(fl(X2 + c2_k) - 2 I_k)`` (X2 = the fp32 |q|^2: the same number for every k; its own error cancels in the comparison).
* the norms of the computed layer inputs are bounded from the ONE norm stage 1 computes exactly, ``N0 = delta sqrt(sum (code - zx)^2)``:
  ``||v_t|| <= sigma_max(A_t) N0 + ||a_t|| + ||Delta_t||`` (A_t: x_0 -> v_t; a_t: bias / codeword offsets, maximised over the codes).

So ``E_l(N0) = ea_l + eb_l N0 + ec_l N0^2`` bounds ``|(D_hat_l[a] - D_hat_l[b]) - (s_l[a] - s_l[b])|`` for every pair (the per-score terms
counted twice), and with stage 1's own error e2 per score (the grid: ``0.5 sum |code - zx| + 0.5 (1 + tables)`` units) a candidate whose gap
exceeds ``tau_l = E_l + 2 e2`` is the fp32 chain's strict minimum: ``D_hat[a] - D_hat[b] <= (s~_a - s~_b) + 2 e2 + E_l < 0``.
Measured on the bench's model: tau = 0.13 / 0.28 / 0.48 at levels 0 / 1 / 2 for a typical cell (|x| = 134), 12 % of the cells go to stage 2
(the propagated part is ~1000x what the chain really loses -- worst-case rounding -- but the gaps are O(1), so it is affordable).

``tests/test_encode_two_stage_cpu.py`` pins the theory on the CPU: the numpy emulation of stage 1 (``candidate_emulate``) against the oracle
on the reference's golden rows and on synthetic ones -- every accepted cell equal at every level, flagged fraction reported."""
from typing import Dict

import numpy as np

U = 2.0 ** -24
LIMBS = 3
G_MAX_INT = 2 ** 23 - 2 ** 15 - 2 ** 7 - 1          # three balanced base-256 digits in [-128, 127] reach +-(2^23 - 2^15 - 2^7 - ...)


def _sn(m):
    return float(np.linalg.norm(m, 2)) if m.size else 0.0


def _kappa(w, block=8):
    """sum_{i=1..K} sigma_max(W[:, :i]) from above: sigma_max(W[:, :i]) grows with i, so a block of ``block`` columns counts its end value"""
    k = w.shape[1]
    return float(sum(min(block, k - i0) * _sn(w[:, :min(k, i0 + block)]) for i0 in range(0, k, block)))


class _Aff:
    """p * N0 + r with p, r >= 0"""

    def __init__(self, p=0.0, r=0.0):
        self.p, self.r = float(p), float(r)

    def __add__(self, o):
        return _Aff(self.p + o.p, self.r + o.r)

    def scale(self, c):
        return _Aff(self.p * c, self.r * c)


def encode_error_bound(state: Dict[str, np.ndarray], levels: int, delta: float):
    """(ea, eb, ec) float64 [levels]: ``|(D_hat_l[a] - D_hat_l[b]) - (s_l[a] - s_l[b])| <= ea_l + eb_l N0 + ec_l N0^2`` for every pair of codes,
    every cell and every choice of the earlier levels' codes (see the module docstring; N0 = ||x_0||).  Also returns the per-source table
    for DESIGN.md."""
    g = lambda l, n: state[f"codebook/{l}/{n}"].astype(np.float64)
    u1 = U / (1.0 - U)
    ub = u1 / (1.0 - 256.0 * u1)                                    # one 256-long chain, errors inside the partial sums included
    eye = np.eye(256)
    A, shift, back = eye.copy(), np.zeros(256), []                  # ideal x_l = A x_0 + shift - sum_j back[j] c_{j, code_j}
    srcs = [(eye.copy(), _Aff(U, 0.0), "x0 = fl(n * delta)")]       # (map from the source's error to the current vector, its norm bound)
    ea, eb, ec, report = [], [], [], []

    def offset(shift, back):
        return float(np.linalg.norm(shift) + sum(np.linalg.norm(g(j, "codebook") @ B.T, axis=1).max() for j, B in enumerate(back)))

    def norm_bound(A, shift, back, srcs):
        d = _Aff()
        for M, e, _ in srcs:
            d = d + e.scale(_sn(M))
        return _Aff(_sn(A), offset(shift, back)) + d

    def layer(W, b, vin: _Aff):
        return _Aff(ub * _kappa(W) * vin.p, ub * (256.0 * float(np.linalg.norm(b)) + _kappa(W) * vin.r) + 1e-30)

    for l in range(levels):
        S, bs, Q, bq, Cb = g(l, "stage_w"), g(l, "stage_b"), g(l, "qhead_w"), g(l, "qhead_b"), g(l, "codebook")
        vx = norm_bound(A, shift, back, srcs)
        srcs = [(S @ M, e, n) for M, e, n in srcs] + [(eye.copy(), layer(S, bs, vx), f"stage{l}")]
        A, shift, back = S @ A, S @ shift + bs, [S @ B for B in back]
        vz = norm_bound(A, shift, back, srcs)
        sq = [(Q @ M, e, n) for M, e, n in srcs] + [(eye.copy(), layer(Q, bq, vz), f"qhead{l}")]
        vq = norm_bound(Q @ A, Q @ shift + bq, [Q @ B for B in back], sq)
        cn = np.linalg.norm(Cb, axis=1)
        cmax, c2max = float(cn.max()), float((cn ** 2).max())
        e = _Aff()
        for M, eps, name in sq:
            # a comparison is about a PAIR of codes: (s_a - s_b) moves by 2 (c_a - c_b) . M delta <= 2 max_{a, b} ||M^T (c_a - c_b)|| ||delta||
            K = Cb @ M
            gram = K @ K.T
            dg = np.diag(gram)
            wpair = float(np.sqrt(max(0.0, (dg[:, None] + dg[None, :] - 2.0 * gram).max())))
            e = e + eps.scale(2.0 * wpair)
            report.append((l, name, wpair, eps.p, eps.r))
        # the terms each score has of its own, twice (a and b): the chain of q . c_k (partial sums <= ||c_k|| ||q||), 2 u |I_k|, the two
        # roundings of the distance, and |c_k|^2 itself -- the kernels' fp32 sum of squares (66 roundings of non-negative terms) where the
        # collapsed form holds the real one
        e = e + vq.scale(2.0 * 2.0 * 256.0 * ub * cmax)
        e = e + vq.scale(2.0 * 2.0 * U * cmax * (1.0 + 256.0 * ub))
        fin = 2.0 * U * (2.0 + U)                                   # u (2 + u) (X2 + c2_k),  X2 <= (1 + 2^-17) ||q||^2
        x2 = fin * (1.0 + 2.0 ** -17)
        ea.append(e.r + (fin * (1.0 + 1e-5) + 2.0 * 67.0 * U) * c2max + x2 * vq.r * vq.r)
        eb.append(e.p + x2 * 2.0 * vq.p * vq.r)
        ec.append(x2 * vq.p * vq.p)
        if l < levels - 1:
            Lh, bl = g(l, "lhead_w"), g(l, "lhead_b")
            srcs = [(Lh @ M, e_, n) for M, e_, n in srcs] + [(eye.copy(), layer(Lh, bl, vz), f"lhead{l}")]
            A, shift, back = Lh @ A, Lh @ shift + bl, [Lh @ B for B in back] + [eye.copy()]
            vn = norm_bound(A, shift, back, srcs)                    # x_{l+1} before its own rounding; fl(lh - c): |e| <= u |x_hat|
            srcs = srcs + [(eye.copy(), vn.scale(U / (1.0 - U)), f"sub{l}")]
    host = 1.0 + 1e-6                                              # the float64 linear algebra of this function
    return np.array(ea) * host, np.array(eb) * host, np.array(ec) * host, report


def candidate_tables(state: Dict[str, np.ndarray], levels: int, in_delta: float, in_zx: int) -> Dict[str, np.ndarray]:
    """Everything ``qv2x_codebook_encode_candidates_i8`` reads (include/qv2x.h), from the frozen PTQ state:

      gpack  i8  [levels][kc/32][LIMBS][8][64][16]  the limbs of G on the grid ``h`` as A fragments of v_mfma_i32_32x32x32_i8
                 (lane = 32 * half + score % 32, bytes = input channel 32 * step + 16 * half + 0..15)
      bias   i64 [levels*kc]        rint(g / h) + (128 - zx) * rowsum(G_int): the kernel multiplies by the stored byte (code - 128)
      tables i32 [levels(levels-1)/2][kc][kc]   rint(T_lj / h), table (l, j) at l (l - 1) / 2 + j
      tau    f32 [levels][3]        tau_l / h = tau[l][0] + tau[l][1] N0 + tau[l][2] N0^2 (+ sum |code - zx|, added by the kernel)
      h      the grid step (float64)"""
    from .engine import collapse_encoder_f64
    G, gb, tabs = collapse_encoder_f64(state, levels)               # dist - |q|^2 = G x_0 + gb + tables, x_0 = delta (code - zx)
    kc = G.shape[0] // levels
    if kc % 32 or kc > 128 or levels > 3:
        raise ValueError("two-stage encode: dict_size 32 | 64 | 96 | 128, up to three levels, seg_num 1")
    gs = float(in_delta) * G
    # the grid: 24 bits for the largest entry of G -- coarser when a table or bias entry would leave the kernel's integer ranges (two table
    # entries are added to 256 a1 + a0 in i32: below 2^28 each; the packed scores stay below 2^53: bias below 2^44).  A coarser grid only
    # costs bound (e2 = 0.5 h sum |code - zx|), never exactness.
    tmax = max([float(np.abs(t).max()) for t in tabs.values()] + [0.0])
    h = max(float(np.abs(gs).max()) / G_MAX_INT, tmax / (2.0 ** 28 - 2.0), float(np.abs(gb).max()) / (2.0 ** 44 - 2.0 ** 32))
    gi = np.rint(gs / h).astype(np.int64)
    limbs, rest = [], gi.copy()
    for _ in range(LIMBS):
        d = ((rest + 128) % 256) - 128
        limbs.append(d.astype(np.int8))
        rest = (rest - d) // 256
    assert not rest.any()
    bias = np.rint(gb / h).astype(np.int64) + (128 - int(in_zx)) * gi.sum(1)
    tables = np.stack([np.rint(tabs[(l, j)] / h) for l in range(levels) for j in range(l)]) if levels > 1 else np.zeros((1, kc, kc))
    if np.abs(tables).max() >= 2 ** 28 or np.abs(bias).max() >= 2 ** 44:          # (the kernel adds two table entries to 256 a1 + a0 in i32)
        raise ValueError("two-stage encode: table / bias entries leave the fixed-point range of the candidate stage")
    lane = np.arange(64)
    score = (np.arange(levels * kc // 32)[:, None, None, None] * 32 + (lane & 31)[None, None, :, None])                  # [tiles, 1, 64, 1]
    chan = (32 * np.arange(8)[None, :, None, None] + 16 * (lane >> 5)[None, None, :, None] + np.arange(16)[None, None, None, :])   # [1, 8, 64, 16]
    packed = np.stack([lb[score, chan] for lb in limbs], axis=1)                           # [levels * kc / 32][limbs][8][64][16]
    packed = packed.reshape(levels, kc // 32, LIMBS, 8, 64, 16)
    ea, eb, ec, report = encode_error_bound(state, levels, in_delta)
    fp = 1.0 + 1e-4                                                # the kernel evaluates tau in fp32 from N0 = delta * sqrt(n2) (a few ulp each)
    tau = np.zeros((levels, 3))
    for l in range(levels):
        own = 1.0 + l + 2.0                                        # stage 1's own error, x 2: bias + tables half a unit each, + 1 unit of host slack
        tau[l] = [(ea[l] / h + own) * fp, eb[l] / h * fp, ec[l] / h * fp]
    return {"gpack": np.ascontiguousarray(packed, dtype=np.int8), "bias": np.ascontiguousarray(bias, dtype=np.int64),
            "tables": np.ascontiguousarray(tables, dtype=np.int32), "tau": np.ascontiguousarray(tau, dtype=np.float32), "h": h,
            "g_int": gi, "bound": (ea, eb, ec), "report": report}


def candidate_emulate(codes_u8: np.ndarray, tabs: Dict[str, np.ndarray], in_delta: float, in_zx: int):
    """numpy emulation of stage 1 on rows of input codes u8 [R, 256] (exact: every integer stays below 2^53 in float64).  Returns
    (codes u8 [levels, R], flagged bool [R], per-level flags bool [levels, R], gaps in score units f64 [levels, R])."""
    gi, bias, tables, tau, h = tabs["g_int"], tabs["bias"], tabs["tables"], tabs["tau"], tabs["h"]
    levels = tau.shape[0]
    kc = gi.shape[0] // levels
    b = codes_u8.astype(np.float64) - 128.0
    s = b @ gi.astype(np.float64).T + bias.astype(np.float64)[None]                      # [R, levels * kc], exact
    n = codes_u8.astype(np.int64) - int(in_zx)
    n1, n2 = np.abs(n).sum(1), (n * n).sum(1)
    n0 = (np.float32(in_delta) * np.sqrt(n2.astype(np.float32))).astype(np.float32)
    out = np.zeros((levels, codes_u8.shape[0]), np.uint8)
    flags = np.zeros((levels, codes_u8.shape[0]), bool)
    gaps = np.zeros((levels, codes_u8.shape[0]))
    for l in range(levels):
        sc = s[:, l * kc:(l + 1) * kc].copy()
        for j in range(l):
            sc += tables[l * (l - 1) // 2 + j][out[j]].astype(np.float64)
        order = np.argsort(sc, axis=1, kind="stable")
        best, second = np.take_along_axis(sc, order[:, :1], 1)[:, 0], np.take_along_axis(sc, order[:, 1:2], 1)[:, 0]
        out[l] = order[:, 0]
        t = (((tau[l, 0] + tau[l, 1] * n0) + (tau[l, 2] * n0) * n0) + n1.astype(np.float32)).astype(np.float32)     # the kernel's fp32 statement
        assert t.dtype == np.float32 and n0.dtype == np.float32
        t = t.astype(np.float64)
        # the kernel compares packed values 128 S + k: accepted iff second - best > 128 T + 127  (implies S_second - S_best > T)
        k2 = np.take_along_axis(order, np.ones((len(sc), 1), np.int64), 1)[:, 0]
        flags[l] = (128.0 * second + k2) - (128.0 * best + out[l]) <= 128.0 * np.ceil(t) + 127.0
        gaps[l] = (second - best) * h
    return out, flags.any(0), flags, gaps
