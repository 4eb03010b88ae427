// f3 (first kernel of the Pyramid fusion row): weighted_fuse of the HEAL Pyramid model
// (opencood/models/fuse_modules/pyramid_fuse.py:17-62) for ONE scale of one scene:
//     warp every agent's feature map AND its occupancy score map into the ego frame (warp_affine_simple: affine_grid + bilinear
//     grid_sample, zeros outside, align_corners = False), set warped scores that are exactly 0 to -inf, softmax over the agents,
//     NaN (every agent masked) -> 0, out = sum_j p_j * feature_j.
// A workgroup owns 16 consecutive ego cells and works in two phases.  (1) thread = cell (16 threads): the float64 -> fp32 sampling grid
// of fuse_att.h, the four taps and their weights, the warped score and the softmax over the agents -- per-cell scalars, parked in LDS
// (tap cells, tap weights, p per agent).  (2) lanes = channels, four per lane, every wave four of the cells: 4 / 2 / 1 cells per step
// for 64 / 128 / >= 256 channels, the per-cell scalars read back as LDS broadcasts.  (The first version ran phase 1 on every lane of a
// wave per cell: 73 us for the 100 x 352 x 64 level, against 4 us of traffic; 64 cells per wave starves the 25 x 88 level of waves.)
// Agent order in the sums = agent index, as torch.sum(dim=0).
#include "fuse_att.h"

namespace qv2x {
namespace {

struct PyrArgs {
    const float* feats; const float* score; const double* pairwise; float* out;
    const int8_t* feats_i8; int ax; float dx;            // I8: padded i8 BEV [agents][h+2][w+2][c] and its quantizer
    int pad;                                             // fp32 variant: 1 = features AND output are padded maps [..][h+2][w+2][c]
    int agents, h, w, c, hw, L, ego;
    double hm, wm, ratio;
};

constexpr int PREC = 9;                                  // per (cell, agent): 4 tap cells, 4 tap weights, p

// I8: the features are the level's activation codes, dequantized at the four taps ((code - zp) * delta, the value the reference's
// fake-quantized map holds)
template <bool I8>
__global__ __launch_bounds__(256) void pyramid_weighted_fuse_kernel(const PyrArgs a) {
    __shared__ float rec[16][MAXA][PREC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cell0 = blockIdx.x * 16;
    if (threadIdx.x < 16) {   // ---- phase 1: thread = cell ------------------------------------------------------------------------
        const int cell = min(cell0 + (int)threadIdx.x, a.hw - 1);
        const int cy = cell / a.w, cx = cell - cy * a.w;
        const double xn = (2.0 * cx + 1.0) / a.w - 1.0, yn = (2.0 * cy + 1.0) / a.h - 1.0;
        float sc[MAXA];
        float smax = -INFINITY;
#pragma unroll
        for (int ag = 0; ag < MAXA; ++ag) {
            sc[ag] = -INFINITY;
            if (ag < a.agents) {
                const double* T = a.pairwise + ((size_t)a.ego * a.L + ag) * 16;
                const double t00 = T[0], t01 = T[1] * a.hm / a.wm, t02 = T[3] / (a.ratio * a.wm) * 2.0;
                const double t10 = T[4] * a.wm / a.hm, t11 = T[5], t12 = T[7] / (a.ratio * a.hm) * 2.0;
                const float gx = (float)(t00 * xn + t01 * yn + t02), gy = (float)(t10 * xn + t11 * yn + t12);
                const float ix = ((gx + 1.0f) * (float)a.w - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)a.h - 1.0f) / 2.0f;
                const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
                const float wt[4] = {(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)};
                const float tx[4] = {x0, x1, x0, x1}, ty[4] = {y0, y0, y1, y1};
                float sv = 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool in = tx[t] >= 0.0f && tx[t] < (float)a.w && ty[t] >= 0.0f && ty[t] < (float)a.h;
                    const int tc = in ? (int)ty[t] * a.w + (int)tx[t] : -1;
                    rec[threadIdx.x][ag][t] = __int_as_float(tc);
                    rec[threadIdx.x][ag][4 + t] = wt[t];
                    if (in) sv += a.score[(size_t)ag * a.hw + tc] * wt[t];              // taps in (nw, ne, sw, se) order, as grid_sample sums them
                }
                sc[ag] = sv == 0.0f ? -INFINITY : sv;                                   // masked_fill_(scores == 0, -inf)
                smax = fmaxf(smax, sc[ag]);
            }
        }
        float den = 0.0f;
#pragma unroll
        for (int ag = 0; ag < MAXA; ++ag)
            if (ag < a.agents) { sc[ag] = expf(sc[ag] - smax); den += sc[ag]; }          // all masked: exp(nan) -> nan, replaced below
#pragma unroll
        for (int ag = 0; ag < MAXA; ++ag)
            if (ag < a.agents) {
                float p = sc[ag] / den;
                rec[threadIdx.x][ag][8] = (p != p) ? 0.0f : p;                          // torch.where(isnan, 0, .)
            }
    }
    __syncthreads();
    // ---- phase 2: four channels per lane; wave w takes cells 4 w .. 4 w + 3 ------------------------------------------------------------
    const int lpc = min(a.c >> 2, 64);                   // lanes per cell (16, 32, 64)
    const int cps = 64 / lpc;                            // cells per step
    const int sub = lane / lpc, l4 = (lane - sub * lpc) * 4;
    for (int step = 0; step < 4 / cps; ++step) {
        const int ci = wave * 4 + step * cps + sub, cell = cell0 + ci;
        if (cell >= a.hw) continue;
        for (int cb = 0; cb < a.c; cb += 256) {
            const int ch = cb + l4;
            if (ch >= a.c) break;
            float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int ag = 0; ag < a.agents; ++ag) {
                const float* r = rec[ci][ag];
                float f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int tc = __float_as_int(r[t]);
                    if (tc >= 0) {
                        const float wt = r[4 + t];
                        float v[4];
                        if (I8) {
                            const int cy2 = tc / a.w, cx2 = tc - cy2 * a.w;
                            const int wv = *(const int*)(a.feats_i8 + ((size_t)(ag * (a.h + 2) + cy2 + 1) * (a.w + 2) + cx2 + 1) * a.c + ch);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (float)(((wv << (24 - 8 * e)) >> 24) + a.ax) * a.dx;
                        } else {
                            size_t at = (size_t)ag * a.hw + tc;
                            if (a.pad) { const int cy2 = tc / a.w, cx2 = tc - cy2 * a.w; at = (size_t)(ag * (a.h + 2) + cy2 + 1) * (a.w + 2) + cx2 + 1; }
                            const v4f fv = *(const v4f*)(a.feats + at * a.c + ch);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fv[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] += v[e] * wt;
                    }
                }
                const float p = r[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += f[e] * p;
            }
            v4f ov = {o[0], o[1], o[2], o[3]};
            size_t oc = cell;
            if (!I8 && a.pad) { const int cy2 = cell / a.w, cx2 = cell - cy2 * a.w; oc = (size_t)(cy2 + 1) * (a.w + 2) + cx2 + 1; }
            *(v4f*)(a.out + oc * a.c + ch) = ov;
        }
    }
}

}  // namespace
}  // namespace qv2x

static int weighted_fuse_f32(const qv2x_fuse_desc* d, int channels, const float* feats, const float* score, const double* pairwise, float* out,
                             void* stream, int padded);

extern "C" int qv2x_pyramid_weighted_fuse_f32(const qv2x_fuse_desc* d, int channels, const float* feats, const float* score, const double* pairwise,
                                              float* out, void* stream) {
    return weighted_fuse_f32(d, channels, feats, score, pairwise, out, stream, 0);
}

extern "C" int qv2x_pyramid_weighted_fuse_f32p(const qv2x_fuse_desc* d, int channels, const float* feats, const float* score, const double* pairwise,
                                               float* out, void* stream) {
    return weighted_fuse_f32(d, channels, feats, score, pairwise, out, stream, 1);
}

static int weighted_fuse_f32(const qv2x_fuse_desc* d, int channels, const float* feats, const float* score, const double* pairwise, float* out,
                             void* stream, int padded) {
    using namespace qv2x;
    if (!d || !feats || !score || !pairwise || !out) return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_f32: null pointer");
    if (d->agents < 1 || d->agents > MAXA || d->max_cav < d->agents || d->ego < 0 || d->ego >= d->agents)
        return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_f32: 1..%d agents, ego inside, max_cav >= agents", MAXA);
    if (d->h <= 0 || d->w <= 0 || channels < 64 || channels % 64) return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_f32: bad sizes (channels %% 64)");
    if (!(d->h_metres > 0) || !(d->w_metres > 0) || !(d->discrete_ratio > 0)) return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_f32: map extent must be positive");
    PyrArgs a;
    a.feats = feats; a.score = score; a.pairwise = pairwise; a.out = out; a.feats_i8 = nullptr; a.ax = 0; a.dx = 0.0f; a.pad = padded;
    a.agents = d->agents; a.h = d->h; a.w = d->w; a.c = channels; a.hw = d->h * d->w; a.L = d->max_cav; a.ego = d->ego;
    a.hm = d->h_metres; a.wm = d->w_metres; a.ratio = d->discrete_ratio;
    pyramid_weighted_fuse_kernel<false><<<(a.hw + 15) / 16, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_pyramid_weighted_fuse_f32 launch");
}

extern "C" int qv2x_pyramid_weighted_fuse_i8(const qv2x_fuse_desc* d, int channels, const int8_t* feats, int in_zx, float in_delta,
                                             const float* score, const double* pairwise, float* out, void* stream) {
    using namespace qv2x;
    if (!d || !feats || !score || !pairwise || !out) return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_i8: null pointer");
    if (d->agents < 1 || d->agents > MAXA || d->max_cav < d->agents || d->ego < 0 || d->ego >= d->agents)
        return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_i8: 1..%d agents, ego inside, max_cav >= agents", MAXA);
    if (d->h <= 0 || d->w <= 0 || channels < 64 || channels % 64) return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_i8: bad sizes (channels %% 64)");
    if (!(d->h_metres > 0) || !(d->w_metres > 0) || !(d->discrete_ratio > 0)) return fail(QV2X_EINVAL, "qv2x_pyramid_weighted_fuse_i8: map extent must be positive");
    PyrArgs a;
    a.feats = nullptr; a.score = score; a.pairwise = pairwise; a.out = out; a.feats_i8 = feats; a.ax = 128 - in_zx; a.dx = in_delta; a.pad = 0;
    a.agents = d->agents; a.h = d->h; a.w = d->w; a.c = channels; a.hw = d->h * d->w; a.L = d->max_cav; a.ego = d->ego;
    a.hm = d->h_metres; a.wm = d->w_metres; a.ratio = d->discrete_ratio;
    pyramid_weighted_fuse_kernel<true><<<(a.hw + 15) / 16, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_pyramid_weighted_fuse_i8 launch");
}
