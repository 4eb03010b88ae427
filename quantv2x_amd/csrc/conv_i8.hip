// a3 / a4 / a5: 3x3 W8A8 convolution as an implicit GEMM on v_mfma_i32_32x32x32_i8 (gfx950).
//
//   M = N*Ho*Wo output pixels, N = Cout, K = G * 9 * C_g walked as (group, kh, kw, 64-channel chunk).
//   A (activations): padded i8 BEV, so a K-chunk of a pixel is 64 contiguous bytes, no bounds checks.
//   B (weights): [Cout][K] i8, K contiguous and in loop order, so the K-chunk offset is linear in the step.
//   Both are staged through LDS in rows of one K chunk (16-byte chunks XOR-swizzled so that the ds_read_b128 fragment
//   reads of a 16-lane group hit 16 distinct slots) by LDS-DMA through a 3- or 4-stage ring (below).
//
// Unsigned x unsigned codes on a signed MFMA (SURVEY.md §7 "hard parts"): with xs = x - 128, ws = w - 128,
// ax = 128 - zx, aw = 128 - zw,
//     sum (x - zx)(w - zw) = sum xs*ws + aw * sum xs + ax * sum ws + K*ax*aw
// sum xs (per pixel window) comes from v_dot4_i32_i8 on the A fragments already in registers; the last two
// terms are the per-(group, co) constant `corr`.  The border of the padded tensor stores zx - 128 = -ax, i.e.
// (x - zx) = 0 there, which is exactly zero padding in the dequantized domain.
//
// Epilogue (spec shared with oracle/qv2x_oracle.c:orc_conv3x3): y = bias + sum_g float(T_g) * scale[g][co]
// (separate mul and add), ReLU, requantize (q_code: equal to IEEE division + rint), store code - 128.
// Wide stride-1 layers (cout % 256 == 0) have their own kernel and weight layout: conv_i8_wide.hip.
#include <cstdlib>

#include "common.h"

namespace qv2x {

struct ConvArgs {
    const int8_t* in; const int8_t* w; const float* scale; const int* corr; const int* aw; const float* bias; int8_t* out;
    int n, hp, wp, cin_total, stride, cout, ngroups;
    int gc0[QV2X_MAX_GROUPS], gc[QV2X_MAX_GROUPS];
    int ho, wo, M, ktot;
    int out_ctotal, out_c0, relu;
    float out_delta, out_zp;
    const void* res; int res_ax; float res_delta;            // EPI 2 | 3: the shortcut of a residual block (qv2x_conv3x3_i8_res)
};

// LDS rows are BK bytes = CH 16-byte chunks; chunk c of row r is stored at c ^ f(r) so that the 16 lanes of a
// ds_read_b128 group (rows {0-3,12-15,20-27}, ... of a 32-row fragment, same chunk) hit 16 distinct 16-byte slots.
template <int BK>
__device__ __forceinline__ int swz(int row, int chunk) {
    constexpr int CH = BK / 16;       // 4, 8 or 16 chunks per row
    return row * BK + ((chunk ^ ((row >> (CH == 4 ? 2 : (CH == 8 ? 1 : 0))) & (CH - 1))) << 4);
}

// ------------------------------------------------------------------------------------------------------------
// The operand tiles are streamed by LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS without a VGPR round trip) through
// an S-stage ring, so several K-chunks are in flight per workgroup and the per-chunk latency is covered (a
// register-staged double buffer, one chunk of prefetch, left ~1 us per chunk exposed on the small layers).  One
// s_barrier per chunk:
//     wait vmcnt((S-2)*LPS)  ->  s_barrier  ->  issue chunk s+S-1 into the stage freed by chunk s-1  ->  MFMAs of chunk s
// The DMA writes LDS lane-linearly (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE
// address and again on the fragment reads (cdna guide rule 21).  All LDS is one array (a second __shared__ object makes
// hipcc drain vmcnt before every ds_read).
// EPI (single group only): 0 = quantize; 2 = + fp32 shortcut [M][cout], 3 = + dequantized shortcut codes (padded i8 BEV [..][cout]),
// then ReLU and the block's quantizer (the end of QuantBasicBlock, quant_block.py:88-96)
template <int BM, int BN, int WM, int WN, int BK, bool MULTI, int S, int MINW, int EPI = 0>
__global__ __launch_bounds__(256, MINW) void conv3x3_i8_dma_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MT = TM / 32, NT = TN / 32;
    constexpr int CH = BK / 16;
    constexpr int IA = BM * BK / 1024, IB = BN * BK / 1024;     // 1 KiB DMA instructions per tile
    constexpr int LPS = (IA + IB) / 4;                          // per wave per chunk
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int NF = MULTI ? 16 : 1;
    static_assert(WM * WN == 4 && (IA + IB) % 4 == 0 && IA % LPS == 0 && S >= 3, "tile shape");

    __shared__ __attribute__((aligned(16))) int8_t lds[S * STAGE + 4 * TM * 4 + BM * 4];
    int* xbuf = (int*)(lds + S * STAGE);
    int* rowoff = xbuf + 4 * TM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // ---- this wave's DMA instructions: ids [wave*LPS, +LPS) over [A blocks | B blocks] ---------------------------
    const bool loads_a = wave * LPS < IA;                        // wave-uniform
    const int8_t* src[LPS];
    int dst[LPS];                                                // LDS byte offset inside a stage (wave-uniform)
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
        const int id = wave * LPS + j;
        const int blk = loads_a ? id : id - IA;
        const int p = blk * 64 + lane, row = p / CH, c = (p % CH) ^ ((row >> (CH == 4 ? 2 : (CH == 8 ? 1 : 0))) & (CH - 1));
        if (loads_a) {
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            const int img = m / (a.ho * a.wo), rem = m - img * (a.ho * a.wo);
            const int yo = rem / a.wo, xo = rem - yo * a.wo;
            src[j] = a.in + ((size_t)(img * a.hp + yo * a.stride) * a.wp + xo * a.stride) * a.cin_total + c * 16;
            dst[j] = blk * 1024;
        } else {
            src[j] = a.w + (size_t)(n0 + row) * a.ktot + c * 16;
            dst[j] = BM * BK + blk * 1024;
        }
    }
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < a.M) {
            const int img = m / (a.ho * a.wo), rem = m - img * (a.ho * a.wo);
            const int yo = rem / a.wo, xo = rem - yo * a.wo;
            off = (img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1;
        }
        rowoff[tid] = off;
    }

    v16i acc[MT][NT];
    float facc[MT][NT][NF];
    int xs[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        xs[i] = 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
            if (MULTI) {
                const float b = a.bias[n0 + wn * TN + j * 32 + (lane & 31)];
#pragma unroll
                for (int r = 0; r < NF; ++r) facc[i][j][r] = b;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // nothing of ours may sit in the VM queue before the ring starts

    // ---- chunk enumeration over (group, tap, 64/128-channel chunk) for the ISSUE side ---------------------------------
    int total = 0;
#pragma unroll
    for (int g = 0; g < QV2X_MAX_GROUPS; ++g) total += (g < (MULTI ? a.ngroups : 1)) ? 9 * (a.gc[g] / BK) : 0;
    int i_g = 0, i_tap = 0, i_cc = 0, i_chunks = a.gc[0] / BK, i_step = 0;
    auto issue = [&]() {                                        // DMA of chunk i_step into stage i_step % S
        const int kh = i_tap >= 6 ? 2 : (i_tap >= 3 ? 1 : 0), kw = i_tap - kh * 3;
        const int off = loads_a ? (kh * a.wp + kw) * a.cin_total + a.gc0[i_g] + i_cc * BK : i_step * BK;
        int8_t* stage = lds + (i_step % S) * STAGE;
#pragma unroll
        for (int j = 0; j < LPS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + off),
                                             (__attribute__((address_space(3))) void*)(stage + dst[j]), 16, 0, 0);
        ++i_step;
        if (++i_cc == i_chunks) { i_cc = 0; if (++i_tap == 9) { i_tap = 0; ++i_g; i_chunks = a.gc[i_g < QV2X_MAX_GROUPS ? i_g : 0] / BK; } }
    };
#pragma unroll
    for (int p = 0; p < S - 1; ++p)
        if (p < total) issue();

    // per-lane fragment read offsets inside a stage (loop invariant; the stage base is an immediate after unrolling by S)
    constexpr int KS = BK / 32;
    int offA[MT][KS], offB[NT][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ch = ks * 2 + (lane >> 5);
#pragma unroll
        for (int i = 0; i < MT; ++i) offA[i][ks] = swz<BK>(wm * TM + i * 32 + (lane & 31), ch);
#pragma unroll
        for (int j = 0; j < NT; ++j) offB[j][ks] = BM * BK + swz<BK>(wn * TN + j * 32 + (lane & 31), ch);
    }

    int g = 0, g_end = 9 * (a.gc[0] / BK);
    auto fold_group = [&]() {
        // window sums of this group to LDS (C-fragment rows differ from A-fragment rows)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int tot = xs[i] + __shfl_xor(xs[i], 32);
            if (lane < 32) xbuf[wave * TM + i * 32 + lane] = tot;
            xs[i] = 0;
        }
        __syncthreads();
        if (MULTI) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = n0 + wn * TN + j * 32 + (lane & 31);
                const int awv = a.aw[co];
                const int cr = a.corr[g * a.cout + co];
                const float sc = a.scale[g * a.cout + co];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int r = 0; r < NF; ++r) {
                        const int T = acc[i][j][r] + awv * xbuf[wave * TM + i * 32 + mfma32_row(r, lane)] + cr;
                        facc[i][j][r] = facc[i][j][r] + (float)T * sc;
                        acc[i][j][r] = 0;
                    }
                }
            }
            __syncthreads();
        }
    };

    for (int base = 0; base < total; base += S) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int step = base + u;
            if (step < total) {
                // chunk `step` has landed once at most (chunks issued after it) * LPS of this wave's DMAs are outstanding
                const int later = total - 1 - step;
                if (later >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * LPS) : "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // every wave's part of chunk `step` is in LDS; stage (step-1)%S is free
                if (i_step < total) issue();
                const int8_t* stg = lds + u * STAGE;            // base % S == 0, so stage == u: a compile-time offset
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    v4i fa[MT], fb[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa[i] = *(const v4i*)(stg + offA[i][ks]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb[j] = *(const v4i*)(stg + offB[j][ks]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) xs[i] = __builtin_amdgcn_sdot4(fa[i][q], 0x01010101, xs[i], false);
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = MULTI ? __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    }
                }
                if (MULTI && step + 1 == g_end) {
                    fold_group();
                    ++g;
                    g_end += 9 * (a.gc[g < QV2X_MAX_GROUPS ? g : 0] / BK);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // ring is idle: its first bytes become the epilogue staging tile

    if (!MULTI) {
        // Single group: the weights were the A operand, so lane l holds ONE pixel (l & 31) of M tile i and, of N tile j, the 16
        // channels 8 (r >> 2) + 4 (l >> 5) + (r & 3): four runs of four consecutive bytes of the output row.  The window sum is
        // the lane's own (no exchange), the per-channel constants come as 16-byte loads, four results are requantized and packed
        // at a time (q_pack4) and a tile is staged with four ds_write_b32 per lane -- ~13 VALU instructions per output where the
        // pixel-per-row form (one ds_write_b8 and the whole coordinate logic per element) spent ~85.
        constexpr int SP = TN + 16;                              // staging row pitch: 2-way bank spread for the dword writes
        static_assert(4 * TM * SP <= S * STAGE, "epilogue staging fits the idle ring");
        int8_t* stage = lds + wave * (TM * SP);                  // this wave's [TM pixels][TN channels]
        const float rd = 1.0f / a.out_delta, lowc = a.relu ? a.out_zp + 8388608.0f : 8388608.0f;    // the ReLU lives in the clamp (q_pack4)
        const int half = lane >> 5, l31 = lane & 31;
        int tot[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) tot[i] = xs[i] + __shfl_xor(xs[i], 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int c_aw[16], c_cr[16];
            float c_sc[16], c_bs[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = n0 + wn * TN + j * 32 + 8 * g + 4 * half;
                const v4i xa = *(const v4i*)(a.aw + c0), xc = *(const v4i*)(a.corr + c0);
                const v4f xsc = *(const v4f*)(a.scale + c0), xb = *(const v4f*)(a.bias + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { c_aw[4 * g + e] = xa[e]; c_cr[4 * g + e] = xc[e]; c_sc[4 * g + e] = xsc[e]; c_bs[4 * g + e] = xb[e]; }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y[4], rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (EPI == 2) {
                        const int mrow = m0 + wm * TM + i * 32 + l31;
                        if (mrow < a.M) {
                            const v4f rv = *(const v4f*)((const float*)a.res + (size_t)mrow * a.cout + n0 + wn * TN + j * 32 + 8 * g + 4 * half);
#pragma unroll
                            for (int e = 0; e < 4; ++e) rs[e] = rv[e];
                        }
                    }
                    if (EPI == 3) {
                        const int off = rowoff[wm * TM + i * 32 + l31];
                        if (off >= 0) {
                            const int rw = *(const int*)((const int8_t*)a.res + (size_t)off * a.cout + n0 + wn * TN + j * 32 + 8 * g + 4 * half);
#pragma unroll
                            for (int e = 0; e < 4; ++e) rs[e] = (float)(((rw << (24 - 8 * e)) >> 24) + a.res_ax) * a.res_delta;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const int T = acc[i][j][r] + __mul24(c_aw[r], tot[i]) + c_cr[r];
                        float yv = c_bs[r] + (float)T * c_sc[r];
                        if (EPI != 0) yv = yv + rs[e];
                        y[e] = yv;
                    }
                    *(int*)(stage + (i * 32 + l31) * SP + j * 32 + 8 * g + 4 * half) = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp, lowc);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        constexpr int CPR = TN / 16;
#pragma unroll
        for (int t = 0; t < (TM * CPR + 63) / 64; ++t) {
            const int id = lane + t * 64, row = id / CPR, chn = id % CPR;
            if (id < TM * CPR) {
                const int off = rowoff[wm * TM + row];
                if (off >= 0)
                    *(v4i*)(a.out + (size_t)off * a.out_ctotal + a.out_c0 + n0 + wn * TN + chn * 16) = *(const v4i*)(stage + row * SP + chn * 16);
            }
        }
        return;
    }

    // several input groups (the shrinker's first layer when it does not take the wide kernel): pixels are the rows of the C
    // fragment and the per-group fp32 fold has already produced y
    int8_t* stage = lds + wave * (32 * TN);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, lane);
                float y = facc[i][j][r % NF];
                if (a.relu) y = fmaxf(y, 0.0f);
                stage[row * TN + j * 32 + (lane & 31)] = (int8_t)((int)q_code_mul(y, a.out_delta, a.out_zp) - 128);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        constexpr int CPR = TN / 16;
#pragma unroll
        for (int t = 0; t < (32 * CPR + 63) / 64; ++t) {
            const int id = lane + t * 64, row = id / CPR, chn = id % CPR;
            if (id < 32 * CPR) {
                const int off = rowoff[wm * TM + i * 32 + row];
                if (off >= 0)
                    *(v4i*)(a.out + (size_t)off * a.out_ctotal + a.out_c0 + n0 + wn * TN + chn * 16) = *(const v4i*)(stage + row * TN + chn * 16);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool MULTI, int S, int MINW, int EPI = 0>
static int launch_dma(const ConvArgs& a, hipStream_t st) {
    dim3 grid((a.M + BM - 1) / BM, a.cout / BN);
    conv3x3_i8_dma_kernel<BM, BN, WM, WN, BK, MULTI, S, MINW, EPI><<<grid, 256, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_conv3x3_i8 launch");
}

}  // namespace qv2x

static int conv3x3_entry(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w, const float* scale, const int32_t* corr, const int32_t* aw,
                         const float* bias, int res_mode, const void* res, int res_zx, float res_delta, int8_t* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->stride != 1 && d->stride != 2))
        return fail(QV2X_EINVAL, "qv2x_conv3x3_i8: bad shape n=%d h=%d w=%d stride=%d", d->n, d->h, d->w, d->stride);
    if (d->ngroups < 1 || d->ngroups > QV2X_MAX_GROUPS) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8: 1..%d input groups", QV2X_MAX_GROUPS);
    if (d->cout % 64 || d->cin_total % 16 || d->out_ctotal < d->out_c0 + d->cout)
        return fail(QV2X_EALIGN, "qv2x_conv3x3_i8: cout %% 64, cin_total %% 16, out channel window");
    if (((uintptr_t)in & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_conv3x3_i8: in / w / out must be 16-byte aligned");
    if (d->out_ctotal % 16 || d->out_c0 % 16) return fail(QV2X_EALIGN, "qv2x_conv3x3_i8: out_ctotal and out_c0 must be multiples of 16");
    ConvArgs a;
    a.res = res; a.res_ax = 128 - res_zx; a.res_delta = res_delta;
    a.in = in; a.w = w; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    a.n = d->n; a.hp = d->h + 2; a.wp = d->w + 2; a.cin_total = d->cin_total; a.stride = d->stride; a.cout = d->cout;
    a.ngroups = d->ngroups;
    a.ktot = 0;
    for (int g = 0; g < QV2X_MAX_GROUPS; ++g) {
        a.gc0[g] = g < d->ngroups ? d->group_c0[g] : 0;
        a.gc[g] = g < d->ngroups ? d->group_c[g] : 0;
        if (g < d->ngroups) {
            if (a.gc[g] <= 0 || a.gc[g] % 64 || a.gc0[g] % 16 || a.gc0[g] + a.gc[g] > d->cin_total)
                return fail(QV2X_EALIGN, "qv2x_conv3x3_i8: group %d (c0=%d, c=%d): channels must come in multiples of 64", g, a.gc0[g], a.gc[g]);
            a.ktot += 9 * a.gc[g];
        }
    }
    a.ho = (d->h + 2 - 3) / d->stride + 1;
    a.wo = (d->w + 2 - 3) / d->stride + 1;
    a.M = d->n * a.ho * a.wo;
    a.out_ctotal = d->out_ctotal; a.out_c0 = d->out_c0; a.relu = d->relu;
    a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    if (!(a.out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8: out_delta must be positive");
    hipStream_t st = (hipStream_t)stream;
    bool k128 = true;
    for (int g = 0; g < a.ngroups; ++g) k128 = k128 && (a.gc[g] % 128 == 0);
    const bool multi = a.ngroups > 1;
    if (res_mode) {
        if (multi || a.cout != 64 || a.gc[0] != 64 || !res || ((uintptr_t)res & 15) || (res_mode != 2 && res_mode != 3))
            return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_res: one input group of 64 channels, cout 64, res_mode 2 | 3, aligned shortcut");
        return res_mode == 2 ? launch_dma<64, 64, 2, 2, 64, false, 3, 4, 2>(a, st) : launch_dma<64, 64, 2, 2, 64, false, 3, 4, 3>(a, st);   // (four workgroups per CU: the shortcut epilogue spilled 104 bytes at six, 28 at five)
    }
#ifndef QV2X_CONV_FORCE
#define QV2X_CONV_FORCE 0     // dev builds: 1 never the 128 x 128 variant, 2 also BK = 128 instead of 256, 3 only BK = 128 instead of 256
#endif
    const bool large = a.cout % 128 == 0 && a.M >= 16384 && QV2X_CONV_FORCE != 1 && QV2X_CONV_FORCE != 2;
    if (multi) {
        if (a.cout % 128) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8: multi-group input needs cout %% 128 == 0");
        return launch_dma<128, 128, 2, 2, 64, true, 4, 2>(a, st);
    }
    // (128-byte K chunks for the 128 x 128 tile -- 96 KB ring, one workgroup per CU -- are slower: 36 vs 25-27 us on the 128 / 256-channel
    //  levels at a batch of 8)
    if (large) return launch_dma<128, 128, 2, 2, 64, false, 3, 3>(a, st);
    // 256-byte K chunks (fewest barriers, 98 KB of LDS: one workgroup per CU) while the grid fits the chip in one round; beyond
    // that two resident workgroups per CU with 128-byte chunks win (25 x 88 x 256 layers: 10.0 vs 10.9 us for one frame, 27.0 vs
    // 21.1 us for a batch of four, profiles/r02_conv_dispatch_ablation.log)
    const long long wgs64 = (long long)((a.M + 63) / 64) * (a.cout / 64);
    if (a.gc[0] % 256 == 0 && (QV2X_CONV_FORCE ? QV2X_CONV_FORCE < 2 : wgs64 <= 256)) return launch_dma<64, 64, 2, 2, 256, false, 3, 1>(a, st);
    // three stages rather than four: the extra resident workgroup (3 or 6 per CU) hides more latency than the fourth stage did
    // (64->64 layer at batch 8: 38.7 -> 32.4 us; 25x88 128-channel layer at batch 4: 20.4 -> 15.7 us)
#if QV2X_CONV_FORCE != 4
    // a 64-channel layer over a large map (the stride-2 first layer of level 0 at a batch): 192-pixel tiles -- three 32 x 32 accumulators
    // per wave and K chunk instead of one against the same weight tile and barrier
    if (!k128 && a.cout == 64 && a.M >= 262144) return launch_dma<192, 64, 2, 2, 64, false, 3, 3>(a, st);
#endif
    return k128 ? launch_dma<64, 64, 2, 2, 128, false, 3, 3>(a, st) : launch_dma<64, 64, 2, 2, 64, false, 3, 6>(a, st);
}

extern "C" int qv2x_conv3x3_i8(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w, const float* scale,
                               const int32_t* corr, const int32_t* aw, const float* bias, int8_t* out, void* stream) {
    return conv3x3_entry(d, in, w, scale, corr, aw, bias, 0, nullptr, 128, 0.0f, out, stream);
}

extern "C" int qv2x_conv3x3_i8_res(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w, const float* scale, const int32_t* corr,
                                   const int32_t* aw, const float* bias, int res_mode, const void* res, int res_zx, float res_delta,
                                   int8_t* out, void* stream) {
    if (res_mode != 2 && res_mode != 3) return qv2x::fail(QV2X_EINVAL, "qv2x_conv3x3_i8_res: res_mode 2 (fp32 shortcut) or 3 (shortcut codes)");
    return conv3x3_entry(d, in, w, scale, corr, aw, bias, res_mode, res, res_zx, res_delta, out, stream);
}
