// a7 + a8 + a9 + a10 for scenes of several agents, round 6: the DECODED SOURCE CELLS ARE SHARED between the ego cells that tap them.
//
// fuse_att.hip gives every ego cell a wave that gathers, per agent, the (up to) four bilinear taps' three table rows each: 12 KiB of
// L1 / L2 gathers per (ego cell, agent), and the kernel runs at the rate the CUs' vector-memory path returns them (11 us per agent and
// V2X-Real scene = ~34 TB/s over the chip: profiles/r05_fuse_by_agents.log).  Under a rigid transform neighbouring ego cells tap the same
// source cells -- a tile of 4 x 8 ego cells touches ~45-60 source cells of an agent, not 128 -- so here a sixteen-wave workgroup owns such a
// tile and, agent by agent, decodes every DISTINCT source cell once into an LDS slot (1 KiB: the 256 fp32 channels, the same
// ((bias + T0) + T1) + T2), then blends each ego cell's taps out of LDS: ~1.5 decodes (4.5 KiB of gathers) per (ego cell, agent) instead
// of 12 KiB, the rest on the LDS path.  The arithmetic and its order are fuse_cell_b3's / fuse_cell_n's -- same tap weights, same
// decode sum, taps blended in tap order, agents combined by fuse_combine -- so the fused map is bit-identical
// (tests/test_hip_fuse_tile.py compares the two forms; the oracle-parity suites run on whichever the entry picks).
//
// Setup, once per tile for all agents (thread = (agent, ego cell, tap): 8 x 32 x 4 = 1024): the tap's source cell and weight; the first
// tap in (cell, tap) order that names a source cell is its LEADER and takes the next slot of its agent (ballot prefix); the others copy
// the leader's slot.  More than 64 distinct source cells of one agent (no rigid transform does that to 32 cells; a caller's scaled one
// could) overflow to the direct gather, tap by tap.  Then per agent: decode (4 slots per wave, their loads requested together) |
// barrier | blend (two ego cells per wave, f[agent] in registers) | barrier.
#include "fuse_att.h"

namespace qv2x {
namespace {

constexpr int TH = 4, TW = 8, TC = TH * TW;         // ego cells per tile
constexpr int CAP = 64;                             // LDS slots per agent (1 KiB each)
constexpr int WAVES = 16;

struct TileLds {
    float4 slot[2][CAP][64];                        // 2 x 64 KiB: agent g is blended out of one buffer while agent g + 1 is decoded into the other
    short slot_row[MAXA][CAP][4];                   // the slot's three table rows (level * kc + code byte), fetched once for all agents
    int2 tap_ws[MAXA][TC * 4];                      // (bilinear weight as bits, 0 = skipped; slot of the tap's source cell, >= CAP: overflow)
    int tap_src[MAXA][TC * 4];                      // source cell, -1 = skipped
    int slot_src[MAXA][CAP];
    int lead0[MAXA];                                // leaders among the agent's first 64 taps
    int nslot[MAXA];
};

template <int NA>
__global__ __launch_bounds__(1024) void fuse_att_tile_kernel(FuseArgs a, const SceneList sl, const int tiles_x, const int tiles_y) {
    __shared__ TileLds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sc = blockIdx.y;
    a.agents = sl.agents[sc];
    if (a.feats) a.feats = (const float4*)((const float*)a.feats + sl.off[sc]);
    else a.codes += sl.off[sc];
    a.pairwise += (size_t)sc * a.L * a.L * 16;
    const int ty0 = ((int)blockIdx.x / tiles_x) * TH, tx0 = ((int)blockIdx.x % tiles_x) * TW;

    // ---- setup: thread (agent, cell, tap) -----------------------------------------------------------------------------------------------------
    const int ag = tid >> 7, i = tid & 127, c = i >> 2, lt = i & 3;
    int src = -1;
    float wt = 0.0f;
    {
        const int cy = ty0 + c / TW, cx = tx0 + c % TW;
        if (ag < a.agents && cy < a.h && cx < a.w) {
            // normalize_pairwise_tfm + affine_grid + the bilinear tap, statement for statement as fuse_cell_b3 (fuse_att.h)
            const double xn = (2.0 * cx + 1.0) / a.w - 1.0, yn = (2.0 * cy + 1.0) / a.h - 1.0;
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
            const float w = wx * wy;
            if (w != 0.0f && tx >= 0.0f && tx < (float)a.w && ty >= 0.0f && ty < (float)a.h) {
                wt = w;
                src = (int)ty * a.w + (int)tx;
            }
        }
    }
    if (ag < NA) s.tap_src[ag][i] = src;
    __syncthreads();
    int leader = i;                                     // the first tap of this agent with the same source cell: all 128 compared, the loads
    if (src >= 0 && ag < NA) {                          // independent of one another (a loop with an early exit waits a whole LDS round trip
        const int4* ts = (const int4*)s.tap_src[ag];    //  per step: 12k cycles per tile in the first version)
#pragma unroll 8
        for (int j4 = 31; j4 >= 0; --j4) {
            const int4 v = ts[j4];
            if (v.w == src) leader = 4 * j4 + 3;
            if (v.z == src) leader = 4 * j4 + 2;
            if (v.y == src) leader = 4 * j4 + 1;
            if (v.x == src) leader = 4 * j4;
        }
    }
    const bool leads = src >= 0 && leader == i && ag < NA;
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(leads);
    const int before = __builtin_popcountll(lm & ((1ull << lane) - 1ull));
    if (ag < NA && (i >> 6) == 0 && lane == 0) s.lead0[ag] = __builtin_popcountll(lm);
    __syncthreads();
    if (ag < NA) {
        if (leads) {
            const int slot = before + ((i >> 6) ? s.lead0[ag] : 0);
            s.tap_ws[ag][i] = make_int2(__builtin_bit_cast(int, wt), slot);
            if (slot < CAP) s.slot_src[ag][slot] = src;
        }
        if ((i >> 6) == 1 && lane == 0) s.nslot[ag] = s.lead0[ag] + __builtin_popcountll(lm);
    }
    __syncthreads();
    if (ag < NA && !leads) s.tap_ws[ag][i] = make_int2(__builtin_bit_cast(int, wt), src >= 0 ? s.tap_ws[ag][leader].y : 0);   // (skipped taps: weight 0, slot 0)

    // the code bytes of every (agent, slot, level) in one round trip: 8 x 64 x 3 loads over the 1024 threads
    const bool coded = !a.feats && a.levels == 3;
    if (coded) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int id = tid + 1024 * e;                          // (agent, slot, level)
            const int g = id / (CAP * 3), r = id - g * (CAP * 3), sl_ = r / 3, lv = r - sl_ * 3;
            if (g < NA && g < a.agents && sl_ < (s.nslot[g] < CAP ? s.nslot[g] : CAP))
                s.slot_row[g][sl_][lv] = (short)(lv * a.kc + a.codes[(size_t)g * a.code_agent_stride + (size_t)lv * a.code_level_stride + s.slot_src[g][sl_]]);
        }
    }
    __syncthreads();

    // ---- agent by agent: decode the distinct source cells (4 per wave, their rows requested together and one agent AHEAD), blend the tile's
    //      ego cells out of LDS.  One barrier per agent: the buffer agent g + 1 is decoded into was last read by the blend of agent g - 1, which
    //      every wave finished before it passed agent g's barrier. --------------------------------------------------------------------------
    constexpr int SPW = CAP / WAVES;                    // slots per wave
    const float4 bias = coded ? a.lut_bias[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r0[SPW], r1[SPW], r2[SPW];
#define QV2X_TILE_REQUEST(G)                                                                                                     \
    {                                                                                                                            \
        const int n_ = s.nslot[G] < CAP ? s.nslot[G] : CAP;                                                                      \
        _Pragma("unroll") for (int q = 0; q < SPW; ++q) {                                                                        \
            const int sl_ = wave + WAVES * q;                                                                                    \
            if (sl_ < n_) {                                                                                                      \
                if (coded) {                                                                                                     \
                    r0[q] = table_row(a, s.slot_row[G][sl_][0], lane);                                                           \
                    r1[q] = table_row(a, s.slot_row[G][sl_][1], lane);                                                           \
                    r2[q] = table_row(a, s.slot_row[G][sl_][2], lane);                                                           \
                } else r0[q] = tap_value(a, G, s.slot_src[G][sl_], lane);                                                        \
            }                                                                                                                    \
        }                                                                                                                        \
    }
    float4 f[2][NA];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int g = 0; g < NA; ++g) f[q][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    QV2X_TILE_REQUEST(0)
#pragma unroll
    for (int g = 0; g < NA; ++g) {
        if (g < a.agents) {
            const int n = s.nslot[g] < CAP ? s.nslot[g] : CAP;
#pragma unroll
            for (int q = 0; q < SPW; ++q) {
                const int sl_ = wave + WAVES * q;
                if (sl_ < n) {
                    float4 v = r0[q];
                    if (coded) {                                     // ((bias + T0) + T1) + T2: tap_value's order
                        v = bias;
                        v.x += r0[q].x; v.y += r0[q].y; v.z += r0[q].z; v.w += r0[q].w;
                        v.x += r1[q].x; v.y += r1[q].y; v.z += r1[q].z; v.w += r1[q].w;
                        v.x += r2[q].x; v.y += r2[q].y; v.z += r2[q].z; v.w += r2[q].w;
                    }
                    s.slot[g & 1][sl_][lane] = v;
                }
            }
            if (g + 1 < NA && g + 1 < a.agents) QV2X_TILE_REQUEST(g + 1)
        }
        // this agent's eight (weight, slot) pairs of the wave's two cells: requested before the barrier, together
        int4 ws[2][2];
        if (g < a.agents) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int4* p = (const int4*)&s.tap_ws[g][(2 * wave + q) * 4];
                ws[q][0] = p[0]; ws[q][1] = p[1];
            }
        }
        __syncthreads();
        if (g < a.agents) {
            if (s.nslot[g] <= CAP) {
                // every tap's slot is in LDS: the eight 1 KiB rows requested together, the weights applied unconditionally -- a skipped tap has
                // weight 0 and slot 0: x * 0 = +-0 (the decoded values are finite) and f + (+-0) = f for every f the sum can hold (it starts at
                // +0), the same bits as not adding at all
#pragma unroll
                for (int q = 0; q < 2; ++q) {                        // (a cell at a time: its four rows in flight together, 16 registers)
                    float4 x[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[t] = s.slot[g & 1][t & 1 ? ws[q][t >> 1].w : ws[q][t >> 1].y][lane];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float w = __builtin_bit_cast(float, t & 1 ? ws[q][t >> 1].z : ws[q][t >> 1].x);
                        f[q][g].x += x[t].x * w; f[q][g].y += x[t].y * w; f[q][g].z += x[t].z * w; f[q][g].w += x[t].w * w;
                    }
                }
            } else {
                // more distinct source cells than slots (not under a rigid transform): taps past the slots are gathered directly
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cc = 2 * wave + q;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float w = __builtin_bit_cast(float, t & 1 ? ws[q][t >> 1].z : ws[q][t >> 1].x);
                        const int sl_ = t & 1 ? ws[q][t >> 1].w : ws[q][t >> 1].y;
                        if (w != 0.0f) {
                            const float4 xx = sl_ < CAP ? s.slot[g & 1][sl_][lane] : tap_value(a, g, s.tap_src[g][cc * 4 + t], lane);
                            f[q][g].x += xx.x * w; f[q][g].y += xx.y * w; f[q][g].z += xx.z * w; f[q][g].w += xx.w * w;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int cc = 2 * wave + q, cy = ty0 + cc / TW, cx = tx0 + cc % TW;
        if (cy < a.h && cx < a.w) a.fused[((size_t)sc * a.hw + (size_t)cy * a.w + cx) * 64 + lane] = fuse_combine<NA>(a, f[q], lane);
    }
}

}  // namespace

// launch for `n_scenes` scenes of up to `most` agents (fuse_att.hip picks between this and the wave-per-cell forms)
int fuse_att_tile_launch(const FuseArgs& a, const SceneList& sl, int n_scenes, int most, hipStream_t st) {
    const int tiles_x = (a.w + TW - 1) / TW, tiles_y = (a.h + TH - 1) / TH;
    const dim3 grid(tiles_x * tiles_y, n_scenes);
    switch (fuse_bound(most)) {
        case 1: case 2: fuse_att_tile_kernel<2><<<grid, 1024, 0, st>>>(a, sl, tiles_x, tiles_y); break;
        case 4: fuse_att_tile_kernel<4><<<grid, 1024, 0, st>>>(a, sl, tiles_x, tiles_y); break;
        default: fuse_att_tile_kernel<MAXA><<<grid, 1024, 0, st>>>(a, sl, tiles_x, tiles_y); break;
    }
    return hip_check(hipGetLastError(), "qv2x_fuse_att_batch_f32 (tile form) launch");
}

}  // namespace qv2x
