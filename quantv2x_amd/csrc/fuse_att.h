// Device side of a7 + a8 + a9 + a10 shared by the standalone kernel (fuse_att.hip) and the fused
// fuse -> heads kernel (heads_f32.hip): decode (three LUT gathers) + bilinear warp into the ego frame + per-cell attention.
#pragma once
#include "common.h"

namespace qv2x {

constexpr int MAXA = 8;

struct FuseArgs {
    const uint8_t* codes; const float4* lut; const float4* lut_bias; const float4* feats; float4* fused;
    const double* pairwise;          // device, [L][L][4][4]; row `ego` is used: T[ego][j] = T_j^-1 T_ego
    int agents, h, w, levels, kc, hw, L, ego;
    int fusion;                      // 0 = AttFusion, 1 = MaxFusion
    long long code_agent_stride, code_level_stride;
    double hm, wm, ratio;            // metres covered by the map (H, W) and discrete_ratio of normalize_pairwise_tfm
};

// one row of the decode table (row = plane * kc + code): 1 KiB, a float4 per lane
__device__ __forceinline__ float4 table_row(const FuseArgs& a, int row, int lane) { return a.lut[(size_t)row * 64 + lane]; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float4 tap_value(const FuseArgs& a, int agent, int cell, int lane) {
    if (a.feats) return a.feats[((size_t)agent * a.hw + cell) * 64 + lane];
    float4 v = a.lut_bias[lane];
    if (a.levels == 3) {
        // three levels (every model of the reference): the three code bytes, then the three table rows, then the sum in level order.  With
        // the run-time level count below every load is conditional and hipcc waits for each before it requests the next: six dependent L2
        // round trips per tap instead of two.
        const uint8_t* cp = a.codes + (size_t)agent * a.code_agent_stride + cell;
        const int c0 = cp[0], c1 = cp[(size_t)a.code_level_stride], c2 = cp[2 * (size_t)a.code_level_stride];
        const float4 t0 = table_row(a, c0, lane);
        const float4 t1 = table_row(a, a.kc + c1, lane);
        const float4 t2 = table_row(a, 2 * a.kc + c2, lane);
        v.x += t0.x; v.y += t0.y; v.z += t0.z; v.w += t0.w;
        v.x += t1.x; v.y += t1.y; v.z += t1.z; v.w += t1.w;
        v.x += t2.x; v.y += t2.y; v.z += t2.z; v.w += t2.w;
        return v;
    }
    for (int l = 0; l < a.levels; ++l) {
        const int code = a.codes[(size_t)agent * a.code_agent_stride + (size_t)l * a.code_level_stride + cell];
        const float4 t = table_row(a, l * a.kc + code, lane);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    return v;
}

// a10 on the warped agents' features of one cell (f[ag]: 4 channels per lane): MaxFusion, or AttFusion's ego row
template <int NA>
__device__ __forceinline__ float4 fuse_combine(const FuseArgs& a, const float4 (&f)[NA], int lane) {
    float score[NA];
#pragma unroll
    for (int ag = 0; ag < NA; ++ag) score[ag] = 0.f;
    if (a.fusion == 1) {              // MaxFusion (fusion_in_one.py:118-121): torch.max over the warped agents, out-of-view agents are zeros
        float4 o = f[0];
#pragma unroll
        for (int ag = 1; ag < NA; ++ag)
            if (ag < a.agents) { o.x = fmaxf(o.x, f[ag].x); o.y = fmaxf(o.y, f[ag].y); o.z = fmaxf(o.z, f[ag].z); o.w = fmaxf(o.w, f[ag].w); }
        return o;
    }
    float4 fq = f[0];                 // the ego's own feature is the query
#pragma unroll
    for (int ag = 1; ag < NA; ++ag)
        if (ag == a.ego) fq = f[ag];
    float smax = -INFINITY;
#pragma unroll
    for (int ag = 0; ag < NA; ++ag)
        if (ag < a.agents) {
            const float part = fq.x * f[ag].x + fq.y * f[ag].y + fq.z * f[ag].z + fq.w * f[ag].w;
            score[ag] = wave_sum(part) / 16.0f;        // sqrt(256)
            smax = fmaxf(smax, score[ag]);
        }
    float den = 0.f;
#pragma unroll
    for (int ag = 0; ag < NA; ++ag)
        if (ag < a.agents) { score[ag] = expf(score[ag] - smax); den += score[ag]; }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ag = 0; ag < NA; ++ag)
        if (ag < a.agents) {
            const float p = score[ag] / den;
            o.x += p * f[ag].x; o.y += p * f[ag].y; o.z += p * f[ag].z; o.w += p * f[ag].w;
        }
    return o;
}

// The fused 256-channel feature of one ego BEV cell; the calling wave holds 4 channels per lane.  NA = compile-time bound on
// the agent count (registers: one float4 per agent).  A bilinear tap of weight exactly 0 is skipped: v * 0 adds +-0 to a
// sum that starts at +0, so the result is the same bit pattern -- and an agent whose grid lands on cell centres (the ego
// itself, T = I) costs one decoded tap instead of four.
template <int NA>
__device__ __forceinline__ float4 fuse_cell_n(const FuseArgs& a, int cell, int lane) {
    const int cy = cell / a.w, cx = cell - cy * a.w;
    const double xn = (2.0 * cx + 1.0) / a.w - 1.0, yn = (2.0 * cy + 1.0) / a.h - 1.0;

    float4 f[NA];
#pragma unroll
    for (int ag = 0; ag < NA; ++ag) {
        f[ag] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ag < a.agents) {
            // normalize_pairwise_tfm (transformation_utils.py:68-92) on T[ego][ag]: rows {0,1} x cols {0,1,3}
            const double* T = a.pairwise + ((size_t)a.ego * a.L + ag) * 16;
            const double t00 = T[0], t01 = T[1] * a.hm / a.wm, t02 = T[3] / (a.ratio * a.wm) * 2.0;
            const double t10 = T[4] * a.wm / a.hm, t11 = T[5], t12 = T[7] / (a.ratio * a.hm) * 2.0;
            const float gx = (float)(t00 * xn + t01 * yn + t02);
            const float gy = (float)(t10 * xn + t11 * yn + t12);
            const float ix = ((gx + 1.0f) * (float)a.w - 1.0f) / 2.0f;
            const float iy = ((gy + 1.0f) * (float)a.h - 1.0f) / 2.0f;
            const float x0 = floorf(ix), y0 = floorf(iy);
            const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
            const float wt[4] = {(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)};
            const float tx[4] = {x0, x1, x0, x1}, ty[4] = {y0, y0, y1, y1};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (wt[t] != 0.0f && tx[t] >= 0.0f && tx[t] < (float)a.w && ty[t] >= 0.0f && ty[t] < (float)a.h) {
                    const float4 v = tap_value(a, ag, (int)ty[t] * a.w + (int)tx[t], lane);
                    f[ag].x += v.x * wt[t]; f[ag].y += v.y * wt[t]; f[ag].z += v.z * wt[t]; f[ag].w += v.w * wt[t];
                }
            }
        }
    }
    return fuse_combine<NA>(a, f, lane);
}

// Round 5: the same cell with its memory round trips BATCHED (three code planes: every model of the reference with seg_num 1).
// fuse_cell_n walks agent by agent and tap by tap: code bytes -> table rows -> next tap, up to 64 dependent L2 round trips per cell at
// eight agents, hidden only by the waves a CU holds.  Here
//   * LANE (agent, tap, level) computes its own tap -- the affine grid of ITS agent, its bilinear weight, its bounds test -- and loads ITS
//     code byte: one pass for all agents (twelve lanes per agent, five agents per pass) instead of one redundant pass per agent, and all
//     code bytes of a cell in ONE round trip;
//   * per agent the (up to) twelve table rows of its four taps are requested together, then summed and weighted in fuse_cell_n's order:
//     v = ((bias + T0[c0]) + T1[c1]) + T2[c2], f += v * w tap by tap -- the same fp32 operations in the same order, so the fused map is
//     bit-identical (tests/test_hip_fuse_heads.py); a tap of weight 0 or outside the agent's map is skipped as before.
template <int NA>
__device__ __forceinline__ float4 fuse_cell_b3(const FuseArgs& a, int cell, int lane) {
    const int cy = cell / a.w, cx = cell - cy * a.w;
    const double xn = (2.0 * cx + 1.0) / a.w - 1.0, yn = (2.0 * cy + 1.0) / a.h - 1.0;
    constexpr int PASSES = (NA + 4) / 5;
    int code[PASSES];                   // this lane's table row (level * kc + code byte)
    float wgt[PASSES];                  // its tap's bilinear weight; 0: skipped (weight 0, outside the map, no such agent)
    const int la = lane / 12, lt = (lane % 12) / 3, ll = lane % 3;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int ag = 5 * p + la;
        float w = 0.0f;
        int row = 0;
        if (lane < 60 && ag < a.agents) {
            const double* T = a.pairwise + ((size_t)a.ego * a.L + ag) * 16;
            const double t00 = T[0], t01 = T[1] * a.hm / a.wm, t02 = T[3] / (a.ratio * a.wm) * 2.0;
            const double t10 = T[4] * a.wm / a.hm, t11 = T[5], t12 = T[7] / (a.ratio * a.hm) * 2.0;
            const float gx = (float)(t00 * xn + t01 * yn + t02);
            const float gy = (float)(t10 * xn + t11 * yn + t12);
            const float ix = ((gx + 1.0f) * (float)a.w - 1.0f) / 2.0f;
            const float iy = ((gy + 1.0f) * (float)a.h - 1.0f) / 2.0f;
            const float x0 = floorf(ix), y0 = floorf(iy);
            const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
            const float wx = (lt & 1) ? (ix - x0) : (x1 - ix), wy = (lt & 2) ? (iy - y0) : (y1 - iy);
            const float tx = (lt & 1) ? x1 : x0, ty = (lt & 2) ? y1 : y0;
            const float wt = wx * wy;
            if (wt != 0.0f && tx >= 0.0f && tx < (float)a.w && ty >= 0.0f && ty < (float)a.h) {
                w = wt;
                const int tc = (int)ty * a.w + (int)tx;
                row = ll * a.kc + a.codes[(size_t)ag * a.code_agent_stride + (size_t)ll * a.code_level_stride + tc];
            }
        }
        code[p] = row; wgt[p] = w;
    }
    const float4 bias = a.lut_bias[lane];
    float4 f[NA];
#pragma unroll
    for (int ag = 0; ag < NA; ++ag) {
        f[ag] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ag < a.agents) {
            constexpr int dummy = 0; (void)dummy;
            const int p = ag / 5, l0 = 12 * (ag % 5);
            float wv[4];
            float4 r0[4], r1[4], r2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wv[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wgt[p]), l0 + 3 * t));
                if (wv[t] != 0.0f) {
                    r0[t] = table_row(a, __builtin_amdgcn_readlane(code[p], l0 + 3 * t), lane);
                    r1[t] = table_row(a, __builtin_amdgcn_readlane(code[p], l0 + 3 * t + 1), lane);
                    r2[t] = table_row(a, __builtin_amdgcn_readlane(code[p], l0 + 3 * t + 2), lane);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (wv[t] != 0.0f) {
                    float4 v = bias;
                    v.x += r0[t].x; v.y += r0[t].y; v.z += r0[t].z; v.w += r0[t].w;
                    v.x += r1[t].x; v.y += r1[t].y; v.z += r1[t].z; v.w += r1[t].w;
                    v.x += r2[t].x; v.y += r2[t].y; v.z += r2[t].z; v.w += r2[t].w;
                    f[ag].x += v.x * wv[t]; f[ag].y += v.y * wv[t]; f[ag].z += v.z * wv[t]; f[ag].w += v.w * wv[t];
                }
            }
        }
    }
    return fuse_combine<NA>(a, f, lane);
}

// several scenes in one launch: scene s = `agents[s]` agents whose data starts `off[s]` bytes into the codes (floats into feats)
constexpr int MAX_SCENES = 64;
struct SceneList { long long off[MAX_SCENES]; int agents[MAX_SCENES]; };

// host: smallest compiled bound that covers `agents`
inline int fuse_bound(int agents) { return agents <= 1 ? 1 : (agents <= 2 ? 2 : (agents <= 4 ? 4 : MAXA)); }

// shared argument checks + FuseArgs from the public descriptor (host)
int fuse_args_from_desc(const qv2x_fuse_desc* d, const uint8_t* codes, const float* lut, const float* lut_bias, const float* feats,
                        const double* pairwise, const char* who, FuseArgs& a);

}  // namespace qv2x
