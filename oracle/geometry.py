"""TEST INFRASTRUCTURE: numpy restatement of the geometric part of the hot path.

  normalize_pairwise_tfm     opencood/utils/transformation_utils.py:68-92
  affine_grid + grid_sample  torch semantics used by warp_affine_simple
                             (opencood/models/sub_modules/torch_transformation_utils.py:323-332):
                             align_corners=False, bilinear, zeros padding
  AttFusion                  opencood/models/fuse_modules/fusion_in_one.py:126-151 (+ SDPA :41-45)
"""
import numpy as np


def normalize_pairwise_tfm(t, H, W, discrete_ratio, downsample_rate=1):
    a = np.array(t[..., [0, 1], :][..., [0, 1, 3]], copy=True)
    a[..., 0, 1] = a[..., 0, 1] * H / W
    a[..., 1, 0] = a[..., 1, 0] * W / H
    a[..., 0, 2] = a[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    a[..., 1, 2] = a[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return a


def affine_grid(theta, h, w):
    """theta [n, 2, 3] (float64 in the reference's data path) -> sampling grid [n, h, w, 2] in theta's dtype."""
    dt = theta.dtype
    xs = ((2 * np.arange(w) + 1) / w - 1).astype(dt)
    ys = ((2 * np.arange(h) + 1) / h - 1).astype(dt)
    gx = theta[:, None, None, 0, 0] * xs[None, None, :] + theta[:, None, None, 0, 1] * ys[None, :, None] + theta[:, None, None, 0, 2]
    gy = theta[:, None, None, 1, 0] * xs[None, None, :] + theta[:, None, None, 1, 1] * ys[None, :, None] + theta[:, None, None, 1, 2]
    return np.stack([gx, gy], axis=-1)


def grid_sample_bilinear_zeros(src, grid):
    """src [n, h, w, c] float32 (channels last), grid [n, ho, wo, 2] float32 -> [n, ho, wo, c]."""
    n, h, w, c = src.shape
    gx, gy = grid[..., 0].astype(np.float32), grid[..., 1].astype(np.float32)
    ix = ((gx + np.float32(1)) * np.float32(w) - np.float32(1)) / np.float32(2)
    iy = ((gy + np.float32(1)) * np.float32(h) - np.float32(1)) / np.float32(2)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = np.zeros(grid.shape[:3] + (c,), dtype=np.float32)
    bi = np.arange(n)[:, None, None]
    for xx, yy, ww in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        xi = np.clip(xx, 0, w - 1).astype(np.int64)
        yi = np.clip(yy, 0, h - 1).astype(np.int64)
        out += src[bi, yi, xi] * (ww * ok).astype(np.float32)[..., None]
    return out


def warp_to_ego(feats, affine_b, n):
    """feats [n, h, w, c]; affine_b [L, L, 2, 3]: every agent resampled into agent 0's frame."""
    theta = affine_b[:n, :n][0]
    grid = affine_grid(theta, feats.shape[1], feats.shape[2]).astype(np.float32)
    return grid_sample_bilinear_zeros(feats, grid)


def att_fuse(warped):
    """warped [n, h, w, c] -> ego row of softmax(x x^T / sqrt(c)) x per cell: [h, w, c]."""
    c = warped.shape[-1]
    score = np.einsum("hwc,nhwc->hwn", warped[0], warped).astype(np.float32) / np.float32(np.sqrt(c))
    score = score - score.max(axis=-1, keepdims=True)
    p = np.exp(score)
    p = p / p.sum(axis=-1, keepdims=True)
    return np.einsum("hwn,nhwc->hwc", p.astype(np.float32), warped).astype(np.float32)


# ---- pairwise transforms from the gathered world poses (the multi-GPU link carries poses, not the pairwise matrix) ----------

def max_fuse(warped):
    """``MaxFusion`` (fusion_in_one.py:118-121): elementwise max over the warped agents (an out-of-view agent is a map of zeros)"""
    return warped.max(axis=0)


def solve4(a, b):
    """X with ``a @ X = b`` for 4 x 4 float64 matrices: Gaussian elimination with partial pivoting (first largest pivot), every
    update a separate multiply and subtract, then back substitution -- the fixed operation order ``qv2x_pairwise_from_poses_f64``
    uses on the device (bit-identical to it; equal to ``np.linalg.solve``, the reference's call at
    ``opencood/utils/transformation_utils.py:60``, up to a few ulp)."""
    a = [[np.float64(v) for v in row] for row in np.asarray(a, dtype=np.float64)]
    b = [[np.float64(v) for v in row] for row in np.asarray(b, dtype=np.float64)]
    for k in range(4):
        p = k
        for r in range(k + 1, 4):
            if abs(a[r][k]) > abs(a[p][k]):
                p = r
        a[k], a[p] = a[p], a[k]
        b[k], b[p] = b[p], b[k]
        for r in range(k + 1, 4):
            m = a[r][k] / a[k][k]
            for c in range(k + 1, 4):
                a[r][c] = a[r][c] - m * a[k][c]
            for c in range(4):
                b[r][c] = b[r][c] - m * b[k][c]
    x = [[np.float64(0.0)] * 4 for _ in range(4)]
    for c in range(4):
        for r in range(3, -1, -1):
            s = b[r][c]
            for q in range(r + 1, 4):
                s = s - a[r][q] * x[q][c]
            x[r][c] = s / a[r][r]
    return np.array(x, dtype=np.float64)


def pairwise_from_poses(poses, max_cav):
    """``T[i, j] = T_j^-1 T_i`` identity-padded to ``[L, L, 4, 4]`` (``get_pairwise_transformation``,
    transformation_utils.py:21-66) from the agents' 4 x 4 world poses."""
    n = len(poses)
    t = np.tile(np.eye(4, dtype=np.float64), (max_cav, max_cav, 1, 1))
    for i in range(n):
        for j in range(n):
            if i != j:
                t[i, j] = solve4(poses[j], poses[i])
    return t


# ---- HEAL Pyramid fusion, one scale ----------------------------------------------------------------

def weighted_fuse(x, score, affine_b, n):
    """``pyramid_fuse.weighted_fuse`` (opencood/models/fuse_modules/pyramid_fuse.py:17-62) for one scene and one scale:
    x [n, h, w, c], score [n, h, w, 1] -> [h, w, c].  Features and scores warped into agent 0's frame; warped scores equal to 0
    masked to -inf; softmax over the agents with NaN (every agent masked) -> 0; score-weighted sum in agent order."""
    theta = affine_b[:n, :n][0]
    grid = affine_grid(theta, x.shape[1], x.shape[2]).astype(np.float32)
    fx = grid_sample_bilinear_zeros(np.ascontiguousarray(x, dtype=np.float32), grid)
    sx = grid_sample_bilinear_zeros(np.ascontiguousarray(score, dtype=np.float32), grid)
    s = np.where(sx == 0, -np.inf, sx).astype(np.float32)
    with np.errstate(invalid="ignore"):
        m = s.max(axis=0, keepdims=True)
        e = np.exp(s - m)
        den = np.zeros_like(e[0])
        for j in range(n):
            den = den + e[j]
        p = e / den
    p = np.where(np.isnan(p), np.float32(0), p).astype(np.float32)
    out = np.zeros(fx.shape[1:], np.float32)
    for j in range(n):
        out = out + fx[j] * p[j]
    return out
