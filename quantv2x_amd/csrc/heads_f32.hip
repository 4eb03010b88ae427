// a11: 1x1 detection heads (cls | reg | dir stacked on the channel axis) on fp32 rows, fake-quant weights,
// per-channel output quantizer.  f32 MFMA fma chains (acc0 = bias), results transposed through LDS so that the
// NCHW store is 128 B contiguous per channel.  Also: decode-only LUT kernel for the *_single heads.
#include "fuse_att.h"

namespace qv2x {

// One workgroup = one 32-row tile (32 x 256 fp32 = 32 KB in LDS) x all <= 96 stacked output channels.  Where the rows
// come from is the template parameter:
//   ROWS_GLOBAL  fp32 rows [R][256] in memory, copied in by LDS-DMA (32 instructions of 1 KiB = one row each;
//                fragment-shaped global loads of the same rows -- 32 cache lines per instruction -- kept the texture
//                addresser busy and ran at 44 us per launch);
//   ROWS_DECODE  each agent's own decoded feature (three LUT gathers) for the *_single heads.
// k-quad q of row r lives in 16-byte slot q ^ r, so the 16 lanes of a ds_read_b128 service group ({0-3,12-15,20-27},
// ...) read 16 distinct slots of the 256-byte bank row.  Wave slot ct < cout_pad/32 owns output columns [32 ct, +32);
// the weights stream from L2 as [64][cout_pad][k0, k2, k1, k3] (one float2 per lane per k-quad, coalesced).  The
// transpose tiles of the NCHW store reuse the row buffer: 32 KB per workgroup, five workgroups per CU.
// (Computing the FUSED rows in here as well -- fuse_cell() for 8 cells per wave, then the heads -- was measured slower than
// the two launches: 62.8 vs 24.0 + 31.7 us at one agent, 198 vs 93 + 32 us at four; the gather-latency-bound fusion wants
// one short-lived wave per cell and many of them per CU, which a 32-row GEMM tile does not give it.)
//   ROWS_FUSE<NA> (round 4) the FUSED rows themselves -- decode + warp + attention (fuse_cell_n, fuse_att.h) for the wave's eight cells --
//                so the 36 MB-per-frame fused map is neither written nor read back.  Rounds 1-2 measured this form at ONE frame (1100
//                workgroups = one round of the chip: a launch lasts as long as one tile's dependent chain, 62.8 against 24.0 + 31.7 us) and
//                dropped it; a batch of 32 frames is 27 rounds, where what counts is throughput and the 2.3 GB of map traffic.  Same
//                arithmetic as the two launches, bit for bit.  `fused_tap` (optional) still receives the rows (debug / parity tests).
enum { ROWS_GLOBAL = 0, ROWS_DECODE = 2, ROWS_FUSE = 3 };

#ifdef QV2X_HEADS_TRACE     // dev build only (s_memtime stamps, round 2): s_memtime stamps of thread 0 of every workgroup
__device__ long long g_heads_trace[32768 * 6];
#define HTRACE(k) do { const int hb_ = ((blockIdx.x >> 3) & 1) * (gridDim.x >> 1) + (blockIdx.x >> 4) * 8 + (blockIdx.x & 7); if (threadIdx.x == 0 && hb_ < 32768) g_heads_trace[hb_ * 6 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HTRACE(k) do { } while (0)
#endif

struct HeadArgs {
    const float* x; int R, hw, cout, cout_pad;
    const float* w; const float* bias; const float* da; const float* za; float* out;
};

template <int SRC, int NA = 1>
__device__ __forceinline__ void rows_heads_tile(const HeadArgs& h, const FuseArgs& fa, const int tm, float* smem, const SceneList* sl = nullptr) {
    float* rows = smem;
    HTRACE(0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nct = h.cout_pad >> 5;
    const int par = lane >> 5;

    if (SRC == ROWS_GLOBAL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = wave * 8 + j;
            const int m = tm * 32 + r;
            const int mc = m < h.R ? m : h.R - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(h.x + (size_t)mc * 256 + ((lane ^ r) << 2)),
                                             (__attribute__((address_space(3))) void*)(rows + r * 256), 16, 0, 0);
        }
    } else if (SRC == ROWS_FUSE) {
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int r = wave * 8 + j;
            int m = tm * 32 + r;
            m = __builtin_amdgcn_readfirstlane(m);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < h.R) {
                const int sc = m / h.hw, cell = m - sc * h.hw;
                FuseArgs a = fa;                               // this row's scene: its agents, where their data starts, its pairwise block
                a.agents = sl->agents[sc];
                if (a.feats) a.feats = (const float4*)((const float*)a.feats + sl->off[sc]);
                else a.codes += sl->off[sc];
                a.pairwise += (size_t)sc * a.L * a.L * 16;
                // (round 5: the batched-round-trip form where fuse_att.hip takes it -- three code planes, more than one agent; same bits)
                if (NA > 1 && !a.feats && a.levels == 3 && a.agents > 1) v = fuse_cell_b3<NA>(a, cell, lane);
                else v = fuse_cell_n<NA>(a, cell, lane);
                if (fa.fused) fa.fused[(size_t)m * 64 + lane] = v;
            }
            *(float4*)(rows + r * 256 + ((lane ^ r) << 2)) = v;
        }
    } else {
        // the wave's eight rows, four at a time: all their code bytes first, then all their table rows, then the sums in level order
        // (tap_value's arithmetic).  Row by row and level by level -- and with a run-time level count, which makes every load
        // conditional and hipcc wait for each code byte before it requests the next -- this was 48 dependent L2 round trips per wave:
        // 20k of the 36k cycles a *_single workgroup lived (s_memtime stamps, round 2).  Three levels (every model of the reference) take
        // the unrolled form; other counts the row-by-row one.
        if (fa.levels == 3) {
            constexpr int LV = 3;
            const float4 bias4 = fa.lut_bias[lane];
#pragma unroll
            for (int j0 = 0; j0 < 8; j0 += 4) {
                int code[4][LV];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = tm * 32 + wave * 8 + j0 + j;
                    const int mc = m < h.R ? m : h.R - 1;
                    const int agent = mc / h.hw, cell = mc - agent * h.hw;
                    const uint8_t* cp = fa.codes + (size_t)agent * fa.code_agent_stride + cell;
#pragma unroll
                    for (int l = 0; l < LV; ++l) code[j][l] = cp[(size_t)l * fa.code_level_stride];
                }
                float4 t[4][LV];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int l = 0; l < LV; ++l) t[j][l] = fa.lut[((size_t)l * fa.kc + code[j][l]) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = wave * 8 + j0 + j;
                    float4 v = bias4;
#pragma unroll
                    for (int l = 0; l < LV; ++l) { v.x += t[j][l].x; v.y += t[j][l].y; v.z += t[j][l].z; v.w += t[j][l].w; }
                    if (tm * 32 + r >= h.R) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(rows + r * 256 + ((lane ^ r) << 2)) = v;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = wave * 8 + j;
                const int m = tm * 32 + r;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < h.R) {
                    const int agent = m / h.hw;
                    o = tap_value(fa, agent, m - agent * h.hw, lane);
                }
                *(float4*)(rows + r * 256 + ((lane ^ r) << 2)) = o;
            }
        }
    }
    // a workgroup's wave i sits on SIMD i: rotate the column tiles over the waves from one row tile to the next, or the
    // SIMD of the spare wave (cout_pad = 96: three column tiles, four waves) never sees an MFMA
    const int slot = (wave + tm) & 3;
    const bool active = slot < nct;
    const int ct = active ? slot : 0;
    const float2* wl = (const float2*)h.w + (size_t)(ct * 32 + (lane & 31)) * 2 + par;
    auto loadB = [&](float2 (&dst)[8], int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) dst[t] = wl[(size_t)(q0 + t) * h.cout_pad * 2];
    };
    float2 wa[8], wb[8];
    v16f acc;
    {
        const float b = h.bias[ct * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
    }
    // Round 5: a last column tile of at most 16 real channels (72 = 2 x 32 + 8: the multi-class heads) runs on v_mfma_f32_16x16x4_f32 -- two
    // 16-row tiles against 16 columns, HALF the 32 x 32 tile's MFMA time; the instruction sums its four k in ascending order onto the
    // accumulator (profiles/r01_mfma_probe.log: bit-exact against the fmaf chain), so the outputs are the 32-wide form's bit for bit.
    const bool tail16 = active && ct == nct - 1 && h.cout - 32 * ct <= 16;
    v4f c16[2];
    if (active && !tail16) loadB(wa, 0);
    __syncthreads();                                // rows (DMA or ds_write) of every wave have landed
    HTRACE(1);

    if (tail16) {
        const int j = lane & 15, kk = lane >> 4;
        const float* wt = h.w + (size_t)(ct * 32 + j) * 4 + (kk == 1 ? 2 : (kk == 2 ? 1 : kk));     // (a quad is stored k0, k2, k1, k3)
        const float bia = h.bias[ct * 32 + j];
        c16[0] = v4f{bia, bia, bia, bia}; c16[1] = c16[0];
        const float* a0p = rows + j * 256 + kk;
        const float* a1p = rows + (16 + j) * 256 + kk;
#pragma unroll 1
        for (int q0 = 0; q0 < 64; q0 += 16) {
            float bw[16], x0[16], x1[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) bw[t] = wt[(size_t)(q0 + t) * h.cout_pad * 4];
#pragma unroll
            for (int t = 0; t < 16; ++t) { x0[t] = a0p[((q0 + t) ^ j) << 2]; x1[t] = a1p[((q0 + t) ^ (16 + j)) << 2]; }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                c16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[t], bw[t], c16[0], 0, 0, 0);
                c16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[t], bw[t], c16[1], 0, 0, 0);
            }
        }
    } else if (active) {
        const int r = lane & 31;
        const float* ar = rows + r * 256;
        auto block = [&](const float2 (&bv)[8], int q0) __attribute__((always_inline)) {
            float a0[8], a1[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float4 av = *(const float4*)(ar + (((q0 + t) ^ r) << 2));
                a0[t] = par ? av.y : av.x; a1[t] = par ? av.w : av.z;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], bv[t].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], bv[t].y, acc, 0, 0, 0);
            }
        };
        for (int q0 = 0; q0 < 64; q0 += 16) {
            loadB(wb, q0 + 8);
            __builtin_amdgcn_sched_barrier(0);
            block(wa, q0);
            __builtin_amdgcn_sched_barrier(0);
            if (q0 + 16 < 64) loadB(wa, q0 + 16);
            __builtin_amdgcn_sched_barrier(0);
            block(wb, q0 + 8);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    HTRACE(2);
    __syncthreads();                                // every wave is done reading the rows: the transpose tiles reuse that LDS
    HTRACE(3);
    if (active) {
        float (*tr)[33] = (float (*)[33])(smem + ct * 32 * 33);
        if (tail16) {                               // D of the 16 x 16 form: lane = column lane & 15, register r = row 4 (lane >> 4) + r
            const int co = ct * 32 + (lane & 15);
            const float d = h.da[co], z = h.za[co];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) {
                    float y = c16[hf][r2];
                    if (d > 0.0f) y = (q_code(y, d, z) - z) * d;
                    tr[16 * hf + 4 * (lane >> 4) + r2][lane & 15] = y;
                }
        } else {
            const int co = ct * 32 + (lane & 31);
            const float d = h.da[co], z = h.za[co];
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                float y = acc[r2];
                if (d > 0.0f) y = (q_code(y, d, z) - z) * d;
                tr[mfma32_row(r2, lane)][lane & 31] = y;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // lane -> cell lane&31 (128 B contiguous per channel in the NCHW output), channels (lane>>5) + 2*c
        const int mm = tm * 32 + (lane & 31);
        const int bi = mm / h.hw, cell = mm - bi * h.hw;
        float* ob = h.out + (size_t)bi * h.cout * h.hw + cell;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ch = ct * 32 + par + 2 * c;
            if (ch < h.cout && mm < h.R) ob[(size_t)ch * h.hw] = tr[lane & 31][par + 2 * c];
        }
    }
    HTRACE(4);
}

template <int SRC>
__global__ __launch_bounds__(256) void rows_heads_kernel(const HeadArgs h, const FuseArgs fa) {
    __shared__ __attribute__((aligned(16))) float smem[32 * 256];
    rows_heads_tile<SRC>(h, fa, blockIdx.x, smem);
}

// a7-a11 in one launch: every 32-cell tile is fused (decode + warp + attention) and multiplied by the heads where it stands
template <int NA>
__global__ __launch_bounds__(256) void fuse_heads_kernel(const HeadArgs h, const FuseArgs fa, const SceneList sl) {
    __shared__ __attribute__((aligned(16))) float smem[32 * 256];
    rows_heads_tile<ROWS_FUSE, NA>(h, fa, blockIdx.x, smem, &sl);
}

// The fused-feature heads (rows from memory) and the *_single heads (rows decoded from the codes) of one frame in ONE
// launch, blockIdx.y = job: two 1100-workgroup grids whose copy-in / GEMM / store phases interleave, one launch gap less.
// (block id = 16 (tile / 8) + 8 job + tile % 8: block ids go round the eight XCDs, so the job bit sits above them -- every XCD and CU gets
// both jobs alternately and the MFMA-bound fused-map tiles run beside the latency-bound decode tiles; as a (tiles, 2) grid every job-0
// workgroup was dispatched before the first job-1 one, and with the job in bit 0 four XCDs would get all the fused-map tiles)
__global__ __launch_bounds__(256) void rows_heads_pair_kernel(const HeadArgs h0, const HeadArgs h1, const FuseArgs fa1) {
    __shared__ __attribute__((aligned(16))) float smem[32 * 256];
    const int tile = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    if (((blockIdx.x >> 3) & 1) == 0) {
        if (tile * 32 < h0.R) rows_heads_tile<ROWS_GLOBAL>(h0, fa1, tile, smem);
    } else {
        if (tile * 32 < h1.R) rows_heads_tile<ROWS_DECODE>(h1, fa1, tile, smem);
    }
}

static int head_args(const char* who, int R, int hw, int cout, int cout_pad, const float* w, const float* bias, const float* da,
                     const float* za, float* out, HeadArgs& h) {
    if (!w || !bias || !da || !za || !out) return fail(QV2X_EINVAL, "%s: null pointer", who);
    if (R <= 0 || hw <= 0 || R % hw || cout <= 0 || cout > cout_pad || cout_pad % 32 || cout_pad > 96)
        return fail(QV2X_EINVAL, "%s: R=%d hw=%d cout=%d cout_pad=%d (cout_pad in {32, 64, 96})", who, R, hw, cout, cout_pad);
    if ((uintptr_t)w & 15) return fail(QV2X_EALIGN, "%s: w must be 16-byte aligned", who);
    h.x = nullptr; h.R = R; h.hw = hw; h.cout = cout; h.cout_pad = cout_pad; h.w = w; h.bias = bias; h.da = da; h.za = za; h.out = out;
    return QV2X_OK;
}

__global__ __launch_bounds__(256) void decode_lut_kernel(const uint8_t* __restrict__ codes, int R, int levels, int kc,
                                                         const float4* __restrict__ lut, const float4* __restrict__ bias,
                                                         float4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    r = __builtin_amdgcn_readfirstlane(r);
    if (r >= R) return;
    float4 v = bias[lane];
    for (int l = 0; l < levels; ++l) {
        const float4 t = lut[((size_t)l * kc + codes[(size_t)l * R + r]) * 64 + lane];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    out[(size_t)r * 64 + lane] = v;
}

// *_preds_single without a GEMM: decode is a sum of per-level table rows (decode_tables, engine.py) and a 1x1 head is linear, so
//     head(decode(c_0, c_1, c_2))[co] = b'[co] + T'_0[c_0][co] + T'_1[c_1][co] + T'_2[c_2][co],   T'_l = T_l W^T, b' = bias_lut W^T + b
// (made on the host in float64, stored fp32).  The launch this replaces gathered 3 KB of decode-table rows per cell from L2 (845 MB per
// batch of eight frames) to multiply them by a 20-column weight matrix on 32 padded MFMA columns.  Here the tables are
// levels x kc x cout floats (30 KB for 3 x 128 x 20: in LDS), a thread owns one cell: 3 code bytes in, cout floats out (NCHW: coalesced
// over the cells of a wave).  The sum is a different fp32 association than decode-then-dot-product: results agree to ~1e-6 relative before
// the head's output quantizer and to the code except at its rounding boundaries (the tests bound that as for the other heads).
__global__ __launch_bounds__(256) void single_heads_lut_kernel(const uint8_t* __restrict__ codes, int R, int hw, int levels, int kc, int cout,
                                                               int groups, const float* __restrict__ tables, const float* __restrict__ bias,
                                                               const float* __restrict__ da, const float* __restrict__ za, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tab[];       // [levels][kc][cout] then bias [cout], da [cout], za [cout]
    const int nt = levels * kc * cout;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) tab[i] = tables[i];
    for (int i = threadIdx.x; i < cout; i += blockDim.x) { tab[nt + i] = bias[i]; tab[nt + cout + i] = da[i]; tab[nt + 2 * cout + i] = za[i]; }
    __syncthreads();
    // a workgroup = `groups` runs of 64 cells; wave w takes the channels c = w, w + 4, ... of every cell of a run (lane = cell: the NCHW
    // stores of a wave are 256 contiguous bytes per channel).  One frame alone gives 550 workgroups -- with one thread per cell and all its
    // channels the launch was 35 workgroups and 37 us.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int g = 0; g < groups; ++g) {
        const int m = (blockIdx.x * groups + g) * 64 + lane;
        if (m >= R) break;
        const int agent = m / hw, cell = m - agent * hw;
        const float* row[4];
        for (int l = 0; l < levels; ++l) row[l] = tab + ((size_t)l * kc + codes[(size_t)l * R + m]) * cout;
        float* ob = out + (size_t)agent * cout * hw + cell;
        for (int c = wv; c < cout; c += 4) {
            float y = tab[nt + c];
            for (int l = 0; l < levels; ++l) y += row[l][c];
            const float d = tab[nt + cout + c], z = tab[nt + 2 * cout + c];
            if (d > 0.0f) y = (q_code(y, d, z) - z) * d;
            ob[(size_t)c * hw] = y;
        }
    }
}

// Round 4: the same identity for EVERY head of a single-agent scene, all channels in one pass.  AttFusion over one agent returns that agent's
// own feature (fusion_in_one.py:131-151 with record_len 1: the softmax of a single score is 1; T[0][0] = I samples every cell at its own
// centre), so cls / reg / dir on the "fused" map AND the *_single heads are 1x1 heads on decode(codes): CT = c0 + c1 stacked channels,
//     y[co] = b'[co] + T'_0[c_0][co] + T'_1[c_1][co] + T'_2[c_2][co]        (level order; tables made in float64 on the host, engine.py)
// -- no 36 MB fused map, no 3 KB of decode gathers per cell, no GEMM.  The tables (levels x kc x CT floats, 141 KB for 3 x 128 x 92) live in
// LDS for the lifetime of a persistent workgroup (one per CU, eight or sixteen waves); row stride ST floats with ST / 4 odd, so the ds_read_b128 of 64
// lanes with 64 different codes spread over the 64 banks.  A wave takes runs of 64 cells (lane = cell: every NCHW store is 256 contiguous
// bytes per channel), four channels per step: three ds_read_b128, twelve adds, four output quantizers, four stores.
__global__ __launch_bounds__(1024, 1) void table_heads_kernel(const uint8_t* __restrict__ codes, int R, int hw, int levels, int kc, int CT, int ST, int c0, int c1,
                                                             const float* __restrict__ tables, const float* __restrict__ bias, const float* __restrict__ da,
                                                             const float* __restrict__ za, float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ __attribute__((aligned(16))) float tab[];       // [levels * kc][ST], then bias [CT4], da [CT4], za [CT4]
    const int CT4 = (CT + 3) & ~3, rows = levels * kc;
    float* cst = tab + (size_t)rows * ST;
    for (int i = threadIdx.x; i < rows * (ST / 4); i += blockDim.x) {
        const int row = i / (ST / 4), q = i - row * (ST / 4);
        v4f v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < CT) v[e] = tables[(size_t)row * CT + 4 * q + e];
        *(v4f*)(tab + (size_t)row * ST + 4 * q) = v;
    }
    for (int i = threadIdx.x; i < CT4; i += blockDim.x) {
        cst[i] = i < CT ? bias[i] : 0.f; cst[CT4 + i] = i < CT ? da[i] : -1.f; cst[2 * CT4 + i] = i < CT ? za[i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int runs = (R + 63) / 64;
    for (int run = blockIdx.x * nw + wave; run < runs; run += gridDim.x * nw) {
        const int m = run * 64 + lane;
        const int mc = m < R ? m : R - 1;
        const int agent = mc / hw, cell = mc - agent * hw;
        const float* rp[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) rp[l] = tab + (size_t)(l < levels ? l * kc + codes[(size_t)l * R + mc] : 0) * ST;
        float* o0 = out0 ? out0 + (size_t)agent * c0 * hw + cell : nullptr;
        float* o1 = out1 ? out1 + (size_t)agent * c1 * hw + cell : nullptr;
        for (int g = 0; g < CT4 / 4; ++g) {
            v4f y = *(const v4f*)(cst + 4 * g);
            const v4f d4 = *(const v4f*)(cst + CT4 + 4 * g), z4 = *(const v4f*)(cst + 2 * CT4 + 4 * g);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                if (l < levels) { const v4f t = *(const v4f*)(rp[l] + 4 * g); y[0] += t[0]; y[1] += t[1]; y[2] += t[2]; y[3] += t[3]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * g + e;
                float v = y[e];
                if (d4[e] > 0.0f) v = (q_code(v, d4[e], z4[e]) - z4[e]) * d4[e];
                if (m < R && c < CT) {
                    if (c < c0) { if (o0) o0[(size_t)c * hw] = v; }
                    else if (o1) o1[(size_t)(c - c0) * hw] = v;
                }
            }
        }
    }
}

// interior of a padded i8 BEV tensor -> fp32 rows [N*H*W][C]: x = (code - zp) * delta  (models without the codebook)
__global__ __launch_bounds__(256) void dequant_i8_kernel(const int8_t* __restrict__ in, int n, int h, int w, int c, int ax, float delta,
                                                         float* __restrict__ out) {
    const size_t total = (size_t)n * h * w * (c / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (c / 4));
        const size_t pix = i / (c / 4);
        const int x = (int)(pix % w), y = (int)((pix / w) % h), img = (int)(pix / ((size_t)w * h));
        const char4 v = *(const char4*)(in + (((size_t)img * (h + 2) + y + 1) * (w + 2) + x + 1) * c + c4 * 4);
        *(float4*)(out + pix * c + c4 * 4) = make_float4((float)((int)v.x + ax) * delta, (float)((int)v.y + ax) * delta,
                                                         (float)((int)v.z + ax) * delta, (float)((int)v.w + ax) * delta);
    }
}

}  // namespace qv2x

extern "C" int qv2x_dequant_i8_f32(const int8_t* in, int n, int h, int w, int c, int zp, float delta, float* out, void* stream) {
    using namespace qv2x;
    if (!in || !out) return fail(QV2X_EINVAL, "qv2x_dequant_i8_f32: null pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 4) return fail(QV2X_EINVAL, "qv2x_dequant_i8_f32: bad shape (c %% 4 == 0)");
    const size_t total = (size_t)n * h * w * (c / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    dequant_i8_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(in, n, h, w, c, 128 - zp, delta, out);
    return hip_check(hipGetLastError(), "qv2x_dequant_i8_f32 launch");
}

extern "C" int qv2x_heads_f32(const float* x, int R, int hw, int cout, int cout_pad, const float* w, const float* bias,
                              const float* da, const float* za, float* out, void* stream) {
    using namespace qv2x;
    HeadArgs h;
    if (!x) return fail(QV2X_EINVAL, "qv2x_heads_f32: null pointer");
    if (int rc = head_args("qv2x_heads_f32", R, hw, cout, cout_pad, w, bias, da, za, out, h)) return rc;
    if ((uintptr_t)x & 15) return fail(QV2X_EALIGN, "qv2x_heads_f32: x must be 16-byte aligned");
    h.x = x;
    rows_heads_kernel<ROWS_GLOBAL><<<(R + 31) / 32, 256, 0, (hipStream_t)stream>>>(h, FuseArgs{});
    return hip_check(hipGetLastError(), "qv2x_heads_f32 launch");
}

extern "C" int qv2x_fuse_heads_batch_f32(const qv2x_fuse_desc* d, int n_scenes, const int64_t* scene_offset, const int32_t* scene_agents,
                                         const uint8_t* codes, const float* lut, const float* lut_bias, const float* feats, const double* pairwise,
                                         int cout, int cout_pad, const float* w, const float* bias, const float* da, const float* za,
                                         float* out, float* fused_tap, void* stream) {
    using namespace qv2x;
    const char* who = "qv2x_fuse_heads_batch_f32";
    if (!scene_offset || !scene_agents || !d) return fail(QV2X_EINVAL, "%s: null pointer", who);
    if (n_scenes < 1 || n_scenes > MAX_SCENES) return fail(QV2X_EINVAL, "%s: 1..%d scenes, got %d", who, MAX_SCENES, n_scenes);
    SceneList sl{};
    int most = 1;
    for (int s = 0; s < n_scenes; ++s) {
        if (scene_offset[s] < 0 || (feats && scene_offset[s] % 4) || scene_agents[s] < 1 || scene_agents[s] > MAXA || scene_agents[s] > d->max_cav || d->ego >= scene_agents[s])
            return fail(QV2X_EINVAL, "%s: scene %d: offset %lld, agents %d (1..%d, <= max_cav, > ego)", who, s, (long long)scene_offset[s], scene_agents[s], MAXA);
        sl.off[s] = scene_offset[s]; sl.agents[s] = scene_agents[s];
        most = scene_agents[s] > most ? scene_agents[s] : most;
    }
    qv2x_fuse_desc d1 = *d;
    d1.agents = most;
    FuseArgs a;
    if (int rc = fuse_args_from_desc(&d1, codes, lut, lut_bias, feats, pairwise, who, a)) return rc;
    a.fused = (float4*)fused_tap;
    HeadArgs h;
    if (int rc = head_args(who, n_scenes * a.hw, a.hw, cout, cout_pad, w, bias, da, za, out, h)) return rc;
    const dim3 grid((h.R + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    switch (fuse_bound(most)) {
        case 1: fuse_heads_kernel<1><<<grid, 256, 0, st>>>(h, a, sl); break;
        case 2: fuse_heads_kernel<2><<<grid, 256, 0, st>>>(h, a, sl); break;
        case 4: fuse_heads_kernel<4><<<grid, 256, 0, st>>>(h, a, sl); break;
        default: fuse_heads_kernel<MAXA><<<grid, 256, 0, st>>>(h, a, sl); break;
    }
    return hip_check(hipGetLastError(), "qv2x_fuse_heads_batch_f32 launch");
}

extern "C" int qv2x_decode_heads_f32(const uint8_t* codes, int R, int hw, int levels, int kc, const float* lut, const float* lut_bias,
                                     int cout, int cout_pad, const float* w, const float* bias, const float* da, const float* za,
                                     float* out, void* stream) {
    using namespace qv2x;
    if (!codes || !lut || !lut_bias) return fail(QV2X_EINVAL, "qv2x_decode_heads_f32: null pointer");
    if (levels < 1 || levels > 16 || kc < 1 || kc > 256) return fail(QV2X_EINVAL, "qv2x_decode_heads_f32: bad sizes (1..16 code planes = levels * seg_num, dict_size <= 256)");
    HeadArgs h;
    if (int rc = head_args("qv2x_decode_heads_f32", R, hw, cout, cout_pad, w, bias, da, za, out, h)) return rc;
    FuseArgs fa{};
    fa.codes = codes; fa.lut = (const float4*)lut; fa.lut_bias = (const float4*)lut_bias; fa.feats = nullptr;
    fa.levels = levels; fa.kc = kc; fa.hw = hw;
    fa.code_agent_stride = hw; fa.code_level_stride = R;           // codes [levels][R], agent-major rows
    rows_heads_kernel<ROWS_DECODE><<<(R + 31) / 32, 256, 0, (hipStream_t)stream>>>(h, fa);
    return hip_check(hipGetLastError(), "qv2x_decode_heads_f32 launch");
}

extern "C" int qv2x_single_heads_lut_f32(const uint8_t* codes, int R, int hw, int levels, int kc, int cout, const float* tables,
                                         const float* bias, const float* da, const float* za, float* out, void* stream) {
    using namespace qv2x;
    if (!codes || !tables || !bias || !da || !za || !out) return fail(QV2X_EINVAL, "qv2x_single_heads_lut_f32: null pointer");
    if (R <= 0 || hw <= 0 || R % hw || levels < 1 || levels > 4 || kc < 1 || kc > 256 || cout < 1 || cout > 96)
        return fail(QV2X_EINVAL, "qv2x_single_heads_lut_f32: R=%d hw=%d levels=%d kc=%d cout=%d", R, hw, levels, kc, cout);
    const size_t lds = ((size_t)levels * kc * cout + 3 * cout) * sizeof(float);
    if (lds > 64 * 1024) return fail(QV2X_EINVAL, "qv2x_single_heads_lut_f32: tables of %zu bytes do not fit the 64 KB of LDS this kernel takes", lds);
    const int runs = (R + 63) / 64;                                   // runs of 64 cells; ~1024 workgroups, each paying the table copy-in once
    const int groups = runs <= 1024 ? 1 : (runs + 1023) / 1024;
    single_heads_lut_kernel<<<(runs + groups - 1) / groups, 256, lds, (hipStream_t)stream>>>(codes, R, hw, levels, kc, cout, groups, tables, bias, da, za, out);
    return hip_check(hipGetLastError(), "qv2x_single_heads_lut_f32 launch");
}

namespace qv2x {
// Round 6: FOUR consecutive cells per lane (hw % 4 == 0: the four share their agent, and an NCHW store is 16 bytes per lane, 1 KB per wave
// and channel instead of 256 B; the code bytes of a plane come as one dword; the channel constants are read once per four cells).  The
// same fp32 operations per output in the same order as table_heads_kernel.
__global__ __launch_bounds__(1024, 1) void table_heads4_kernel(const uint8_t* __restrict__ codes, int R, int hw, int levels, int kc, int CT, int ST, int c0, int c1,
                                                              const float* __restrict__ tables, const float* __restrict__ bias, const float* __restrict__ da,
                                                              const float* __restrict__ za, float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ __attribute__((aligned(16))) float tab[];       // [levels * kc][ST], then bias [CT4], da [CT4], za [CT4]
    const int CT4 = (CT + 3) & ~3, rows = levels * kc;
    float* cst = tab + (size_t)rows * ST;
    for (int i = threadIdx.x; i < rows * (ST / 4); i += blockDim.x) {
        const int row = i / (ST / 4), q = i - row * (ST / 4);
        v4f v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < CT) v[e] = tables[(size_t)row * CT + 4 * q + e];
        *(v4f*)(tab + (size_t)row * ST + 4 * q) = v;
    }
    for (int i = threadIdx.x; i < CT4; i += blockDim.x) {
        cst[i] = i < CT ? bias[i] : 0.f; cst[CT4 + i] = i < CT ? da[i] : -1.f; cst[2 * CT4 + i] = i < CT ? za[i] : 0.f;
        cst[3 * CT4 + i] = i < CT && da[i] > 0.0f ? 1.0f / da[i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int runs = (R + 255) / 256;
    for (int run = blockIdx.x * nw + wave; run < runs; run += gridDim.x * nw) {
        const int m = run * 256 + 4 * lane;                           // R % 4 == 0: the four cells are all inside or all outside
        const int mc = m < R ? m : R - 4;
        const int agent = mc / hw, cell = mc - agent * hw;
        const float* rp[4][4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const unsigned cw = l < levels ? *(const unsigned*)(codes + (size_t)l * R + mc) : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) rp[l][j] = tab + (size_t)(l < levels ? l * kc + ((cw >> (8 * j)) & 0xff) : 0) * ST;
        }
        float* o0 = out0 ? out0 + (size_t)agent * c0 * hw + cell : nullptr;
        float* o1 = out1 ? out1 + (size_t)agent * c1 * hw + cell : nullptr;
        for (int g = 0; g < CT4 / 4; ++g) {
            const v4f b4 = *(const v4f*)(cst + 4 * g), d4 = *(const v4f*)(cst + CT4 + 4 * g), z4 = *(const v4f*)(cst + 2 * CT4 + 4 * g);
            const v4f r4 = *(const v4f*)(cst + 3 * CT4 + 4 * g);
            v4f y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j] = b4;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    if (l < levels) { const v4f t = *(const v4f*)(rp[l][j] + 4 * g); y[j][0] += t[0]; y[j][1] += t[1]; y[j][2] += t[2]; y[j][3] += t[3]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * g + e;
                v4f v = {y[0][e], y[1][e], y[2][e], y[3][e]};
                if (d4[e] > 0.0f) {                                   // (uniform) clamp(rint(v / d) + z, 0, 255) bit for bit: common.h's division-exact sandwich
                    const unsigned cw = (unsigned)q_pack4_div(v[0], v[1], v[2], v[3], d4[e], r4[e], z4[e]) ^ 0x80808080u;
                    v[0] = ((float)(cw & 0xff) - z4[e]) * d4[e];
                    v[1] = ((float)((cw >> 8) & 0xff) - z4[e]) * d4[e];
                    v[2] = ((float)((cw >> 16) & 0xff) - z4[e]) * d4[e];
                    v[3] = ((float)(cw >> 24) - z4[e]) * d4[e];
                }
                if (m < R && c < CT) {
                    if (c < c0) { if (o0) *(v4f*)(o0 + (size_t)c * hw) = v; }
                    else if (o1) *(v4f*)(o1 + (size_t)(c - c0) * hw) = v;
                }
            }
        }
    }
}

// Round 5: the same look-up for tables that do NOT fit the LDS (seg_num 2 x dict_size 256: six planes of 256 rows x 92 channels = 565 KB;
// three planes of 256 rows: 283 KB) -- the rows stay in global memory (they live in L2) and are fetched WHOLE: lanes = channels (a float4
// per lane, two cells per instruction -- one per half-wave), so a row is one contiguous 368-byte request instead of 64 lanes picking 16
// bytes out of 64 different rows.  A wave takes 32 cells: their code bytes up front (lane = cell), the sums in plane order with the
// channel's quantizer in registers, the results through a [channel][cell] tile of its own LDS into NCHW stores of 128 contiguous bytes.
// The same fp32 operations in the same order as table_heads_kernel.  CT % 4 == 0, planes <= 16.
constexpr int TG_CELLS = 32, TG_PITCH = 33;
__global__ __launch_bounds__(256) void table_heads_global_kernel(const uint8_t* __restrict__ codes, int R, int hw, int planes, int kc, int CT, int c0, int c1,
                                                                 const float* __restrict__ tables, const float* __restrict__ bias, const float* __restrict__ da,
                                                                 const float* __restrict__ za, float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ __attribute__((aligned(16))) float tg[];       // [4 waves][CT][TG_PITCH]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    float* tr = tg + (size_t)wave * CT * TG_PITCH;
    const int run = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    const int m0 = run * TG_CELLS;
    if (m0 >= R) return;
    // this lane's cell (lanes 0-31; the upper half-wave mirrors it) and its code bytes
    const int mc = m0 + l31 < R ? m0 + l31 : R - 1;
    int code[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) code[p] = p < planes ? (int)codes[(size_t)p * R + mc] : 0;
    const int ch = 4 * l31;                                           // this lane's four channels
    const bool chan = ch < CT;
    v4f b4 = {0.f, 0.f, 0.f, 0.f}, d4 = {-1.f, -1.f, -1.f, -1.f}, z4 = {0.f, 0.f, 0.f, 0.f};
    if (chan) { b4 = *(const v4f*)(bias + ch); d4 = *(const v4f*)(da + ch); z4 = *(const v4f*)(za + ch); }
    const float* tcol = tables + (chan ? ch : 0);
#pragma unroll 2
    for (int it = 0; it < TG_CELLS / 2; ++it) {                      // cells 2 it (lower half-wave) and 2 it + 1 (upper)
        const int cell = 2 * it + half;
        v4f t[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (p < planes) {                                         // (uniform)
                const int lo = __builtin_amdgcn_readlane(code[p], 2 * it), hi = __builtin_amdgcn_readlane(code[p], 2 * it + 1);
                t[p] = *(const v4f*)(tcol + (size_t)(p * kc + (half ? hi : lo)) * CT);
            }
        }
        v4f y = b4;
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (p < planes) y += t[p];
        if (chan) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = y[e];
                if (d4[e] > 0.0f) v = (q_code(v, d4[e], z4[e]) - z4[e]) * d4[e];
                tr[(ch + e) * TG_PITCH + cell] = v;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // lane -> cell lane & 31; half-wave h stores the channels h, h + 2, ...
    const int mm = m0 + l31;
    if (mm < R) {
        const int agent = mm / hw, cl = mm - agent * hw;
        float* o0 = out0 ? out0 + (size_t)agent * c0 * hw + cl : nullptr;
        float* o1 = out1 ? out1 + (size_t)agent * c1 * hw + cl : nullptr;
        for (int c = half; c < CT; c += 2) {
            const float v = tr[c * TG_PITCH + l31];
            if (c < c0) o0[(size_t)c * hw] = v; else o1[(size_t)(c - c0) * hw] = v;
        }
    }
}
}  // namespace qv2x

extern "C" int qv2x_table_heads_f32(const uint8_t* codes, int R, int hw, int levels, int kc, int c0, int c1, const float* tables,
                                    const float* bias, const float* da, const float* za, float* out0, float* out1, void* stream) {
    using namespace qv2x;
    const char* who = "qv2x_table_heads_f32";
    if (!codes || !tables || !bias || !da || !za || (!out0 && !out1)) return fail(QV2X_EINVAL, "%s: null pointer", who);
    const int CT = c0 + c1;
    if (R <= 0 || hw <= 0 || R % hw || levels < 1 || levels > 16 || kc < 1 || kc > 256 || c0 < 0 || c1 < 0 || CT < 1 || (c0 > 0) != (out0 != nullptr) || (c1 > 0) != (out1 != nullptr))
        return fail(QV2X_EINVAL, "%s: R=%d hw=%d levels=%d kc=%d c0=%d c1=%d (1..16 planes; an output per non-empty channel set)", who, R, hw, levels, kc, c0, c1);
    const int CT4 = (CT + 3) & ~3;
    const int ST = (CT4 / 4) % 2 ? CT4 : CT4 + 4;                     // row stride in floats, ST / 4 odd: 64 different rows start in 16 different bank quads
    const size_t lds = ((size_t)levels * kc * ST + 4 * CT4) * sizeof(float);
    if (levels > 4 || lds > 160 * 1024) {                             // the tables stay in global memory (round 5)
        if (CT % 4 || CT > 128 || ((uintptr_t)tables & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)da & 15) || ((uintptr_t)za & 15))
            return fail(QV2X_EINVAL, "%s: tables past the 160 KB of LDS need c0 + c1 %% 4 == 0, <= 128, and 16-byte aligned arrays", who);
        const size_t tl = (size_t)4 * CT * TG_PITCH * sizeof(float);
        if (int rc = hip_check(hipFuncSetAttribute((const void*)table_heads_global_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl), who)) return rc;
        const int runs = (R + TG_CELLS - 1) / TG_CELLS;
        table_heads_global_kernel<<<(runs + 3) / 4, 256, tl, (hipStream_t)stream>>>(codes, R, hw, levels, kc, CT, c0, c1, tables, bias, da, za, out0, out1);
        return hip_check(hipGetLastError(), "qv2x_table_heads_f32 launch");
    }
    if (int rc = hip_check(hipFuncSetAttribute((const void*)table_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), who)) return rc;
    int dev = 0, cus = 256, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    // sixteen waves per workgroup (four per SIMD: the kernel is bound by instruction issue -- 153 against 206 us per batch of 32 frames with
    // eight) once every wave has a run of 64 cells to take; below that (one frame: 550 runs) eight, whose table copy-in is over sooner
    int cells_per_lane = 4;
#ifdef QV2X_DEV_KNOBS                                                  // dev builds only: 1 = the one-cell-per-lane form at every size
    static const int cells_env = getenv("QV2X_TABLE_HEADS_CELLS") ? atoi(getenv("QV2X_TABLE_HEADS_CELLS")) : 4;
    cells_per_lane = cells_env;
#endif
    if (cells_per_lane == 4 && hw % 4 == 0 && (R + 255) / 256 >= 16 * cus && !((uintptr_t)codes & 3) && !((uintptr_t)out0 & 15) && !((uintptr_t)out1 & 15)) {
        if (int rc = hip_check(hipFuncSetAttribute((const void*)table_heads4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), who)) return rc;
        table_heads4_kernel<<<cus, 1024, lds, (hipStream_t)stream>>>(codes, R, hw, levels, kc, CT, ST, c0, c1, tables, bias, da, za, out0, out1);
        return hip_check(hipGetLastError(), "qv2x_table_heads_f32 launch");
    }
    const int runs = (R + 63) / 64;
    const int nwav = runs >= 16 * cus ? 16 : 8, want = (runs + nwav - 1) / nwav;
    table_heads_kernel<<<want < cus ? want : cus, nwav * 64, lds, (hipStream_t)stream>>>(codes, R, hw, levels, kc, CT, ST, c0, c1, tables, bias, da, za, out0, out1);
    return hip_check(hipGetLastError(), "qv2x_table_heads_f32 launch");
}

extern "C" int qv2x_decode_lut_f32(const uint8_t* codes, int R, int levels, int kc, const float* lut, const float* lut_bias,
                                   float* out, void* stream) {
    using namespace qv2x;
    if (!codes || !lut || !lut_bias || !out) return fail(QV2X_EINVAL, "qv2x_decode_lut_f32: null pointer");
    if (R <= 0 || levels < 1 || levels > 16 || kc < 1 || kc > 256) return fail(QV2X_EINVAL, "qv2x_decode_lut_f32: bad sizes (1..16 code planes = levels * seg_num, dict_size <= 256)");
    decode_lut_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(codes, R, levels, kc, (const float4*)lut, (const float4*)lut_bias, (float4*)out);
    return hip_check(hipGetLastError(), "qv2x_decode_lut_f32 launch");
}

#ifdef QV2X_HEADS_TRACE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_heads_trace(long long* host_out, int nblocks) {
    using namespace qv2x;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_heads_trace), (size_t)nblocks * 6 * sizeof(long long));
}
#endif

extern "C" int qv2x_heads_pair_f32(const float* x, int R, int hw, int cout, int cout_pad, const float* w, const float* bias,
                                   const float* da, const float* za, float* out,
                                   const uint8_t* codes, int R1, int levels, int kc, const float* lut, const float* lut_bias,
                                   int cout1, int cout_pad1, const float* w1, const float* bias1, const float* da1,
                                   const float* za1, float* out1, void* stream) {
    using namespace qv2x;
    HeadArgs h0, h1;
    if (!x || !codes || !lut || !lut_bias) return fail(QV2X_EINVAL, "qv2x_heads_pair_f32: null pointer");
    if (int rc = head_args("qv2x_heads_pair_f32", R, hw, cout, cout_pad, w, bias, da, za, out, h0)) return rc;
    if (int rc = head_args("qv2x_heads_pair_f32", R1, hw, cout1, cout_pad1, w1, bias1, da1, za1, out1, h1)) return rc;
    if ((uintptr_t)x & 15) return fail(QV2X_EALIGN, "qv2x_heads_pair_f32: x must be 16-byte aligned");
    if (levels < 1 || levels > 16 || kc < 1 || kc > 256) return fail(QV2X_EINVAL, "qv2x_heads_pair_f32: bad sizes (1..16 code planes = levels * seg_num, dict_size <= 256)");
    h0.x = x;
    FuseArgs fa{};
    fa.codes = codes; fa.lut = (const float4*)lut; fa.lut_bias = (const float4*)lut_bias; fa.feats = nullptr;
    fa.levels = levels; fa.kc = kc; fa.hw = hw;
    fa.code_agent_stride = hw; fa.code_level_stride = R1;
    const int tiles = ((R > R1 ? R : R1) + 31) / 32;
    rows_heads_pair_kernel<<<dim3(16 * ((tiles + 7) / 8)), 256, 0, (hipStream_t)stream>>>(h0, h1, fa);
    return hip_check(hipGetLastError(), "qv2x_heads_pair_f32 launch");
}
