// a3 deblocks: ConvTranspose2d(kernel == stride == s) under QuantModel, on v_mfma_f32_32x32x2_f32.
//
// The reference's per-dim-0 weight scale lands on C_in for a [Cin, Cout, s, s] weight (quant_layer.py:192-195),
// i.e. on the reduction axis, so no integer accumulation can be factored; the sum is an ascending-ci fp32 fma
// chain, which is exactly what the f32 MFMA computes (k ascending, C-in first; verified bit-exact against fmaf
// on hardware, tools/probes/mfma_probe.hip).  GEMM view: rows = input pixels, cols = (i*s + j)*Cout + co, K = Cin.
// One wave = 32 pixels x 32 columns; the pixels converted on the fly from the i8 BEV, the weights [Cin/4][cols][4] fp32 from L2.
// The WEIGHTS are the A operand (out^T = W^T x^T, the same ascending-ci chain per output): a lane then holds one pixel and 16
// output channels in four runs of four, which requantize four at a time (q_pack4) and leave as 16-byte stores through a small
// LDS stage -- instead of sixteen 1-byte global stores per lane.
#include "common.h"

#include <cstdlib>

namespace qv2x {

struct DeconvArgs {
    const int8_t* in; const float* w; const float* bias; int8_t* out;
    int n, h, wd, cin, cout, s, ax, ncols, M, relu, out_ctotal, out_c0;
    int np;                                   // pixel-stationary form: pairs of 32-column tiles per item
    float dx, out_delta, out_zp;
};

// one wave tile (`tile` is wave-uniform); stagebuf: this wave's 32 x 48 bytes of LDS
constexpr int DSP = 48;                       // staging pitch per pixel: 32 channel bytes + 16 (2-way bank spread of the dword writes)

// F32IN: the input is an fp32 map [M][cin] (a fused pyramid level, the decoded feature) instead of codes
template <int NT, bool F32IN = false, int MT = 1>
__device__ __forceinline__ void deconv_tile(const DeconvArgs& a, const int tile, int8_t (*stagebuf)[32 * DSP]) {
    const int lane = threadIdx.x & 63;
    const int tiles_n = a.ncols / (32 * NT);
    const int tiles_m = (a.M + 32 * MT - 1) / (32 * MT);
    if (tile >= tiles_m * tiles_n) return;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int par = lane >> 5;

    // MT pixel tiles of 32 per wave: a weight fragment loaded once feeds MT MFMAs (the fp32 weights are 4x the bytes of the int8 pixels)
    const int8_t* src[MT];
    const float* srcf[MT];
    int pixbase[MT];            // padded output pixel index of this lane's pixel at sub-position (i = 0, j = 0); -1 past the end
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m_raw = (tm * MT + i) * 32 + (lane & 31);
        const int m = m_raw < a.M ? m_raw : a.M - 1;
        const int img = m / (a.h * a.wd), rem = m - img * (a.h * a.wd);
        const int y = rem / a.wd, x = rem - y * a.wd;
        src[i] = a.in + ((size_t)(img * (a.h + 2) + y + 1) * (a.wd + 2) + x + 1) * a.cin;
        srcf[i] = (const float*)a.in + (size_t)m * a.cin;
        pixbase[i] = m_raw < a.M ? (img * (a.h * a.s + 2) + y * a.s + 1) * (a.wd * a.s + 2) + x * a.s + 1 : -1;
    }
    const int col0 = tn * (32 * NT) + (lane & 31);
    // weights: [Cin/4][cols][k0, k2, k1, k3]; the half-wave of MFMA k-parity `par` reads one float2 = (k_par, k_par+2)
    const float2* wq = (const float2*)a.w + (size_t)col0 * 2 + par;

    v16f acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

    // one step = 16 input channels = 8 MFMA k-steps x NT column tiles x MT pixel tiles; the next step's operands are requested before
    // the MFMA block of the current one (register double buffer, pinned with sched_barrier: hipcc sinks loads otherwise)
    struct Raw { v4i q; v4f f[4]; };
    auto loadA = [&](Raw (&r)[MT], int k0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (F32IN) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[i].f[q] = *(const v4f*)(srcf[i] + k0 + 4 * q);
            } else {
                r[i].q = *(const v4i*)(src[i] + k0);
            }
        }
    };
    auto loadB = [&](float2 (&dst)[4][NT], int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[q][t] = wq[((size_t)((k0 >> 2) + q) * a.ncols + t * 32) * 2];
    };
    const int sh = par * 8;
    auto step = [&](const Raw (&raw)[MT], const float2 (&b)[4][NT]) {
        float av[MT][8];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {          // MFMA k-step j: k = k0 + 2j + par  ->  byte 2*(j&1) + par of word j>>1
                if (F32IN) {
                    const float e0 = raw[i].f[j >> 1][2 * (j & 1)], e1 = raw[i].f[j >> 1][2 * (j & 1) + 1];
                    av[i][j] = par ? e1 : e0;
                } else {
                    const int xs = (raw[i].q[j >> 1] << (24 - ((j & 1) * 16 + sh))) >> 24;
                    av[i][j] = (float)(xs + a.ax) * a.dx;
                }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32((j & 1) ? b[j >> 1][t].y : b[j >> 1][t].x, av[i][j], acc[i][t], 0, 0, 0);
    };
    Raw r0[MT], r1[MT];
    float2 b0[4][NT], b1[4][NT];
    loadA(r0, 0);
    loadB(b0, 0);
    for (int k0 = 0; k0 < a.cin; k0 += 32) {
        const bool two = k0 + 16 < a.cin;
        if (two) { loadA(r1, k0 + 16); loadB(b1, k0 + 16); }
        __builtin_amdgcn_sched_barrier(0);
        step(r0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (two) {
            if (k0 + 32 < a.cin) { loadA(r0, k0 + 32); loadB(b0, k0 + 32); }
            __builtin_amdgcn_sched_barrier(0);
            step(r1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // lane l now holds pixel (l & 31) and, of column tile t, the 16 columns 8 (r >> 2) + 4 (l >> 5) + (r & 3); Cout is a multiple
    // of 32, so the tile's 32 columns are 32 consecutive output channels of ONE sub-position (di, dj)
    int8_t* stage = stagebuf[threadIdx.x >> 6];
    const int orow = a.wd * a.s + 2;             // output pixels per padded row
    const float rd = 1.0f / a.out_delta, lowc = a.relu ? a.out_zp + 8388608.0f : 8388608.0f;    // the ReLU lives in the clamp (q_pack4)
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pb = __shfl(pixbase[i], lane >> 1);   // the copy-out below moves 16 bytes per lane: pixel lane >> 1, chunk lane & 1
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = tn * (32 * NT) + t * 32;
            const int ij = col / a.cout, co0 = col - ij * a.cout;
            const int di = ij / a.s, dj = ij - di * a.s;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f b4 = *(const v4f*)(a.bias + co0 + 8 * g + 4 * half);
                float yv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[e] = acc[i][t][4 * g + e] + b4[e];
                *(int*)(stage + l31 * DSP + 8 * g + 4 * half) = q_pack4(yv[0], yv[1], yv[2], yv[3], a.out_delta, rd, a.out_zp, lowc);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (pb >= 0)
                *(v4i*)(a.out + (size_t)(pb + di * orow + dj) * a.out_ctotal + a.out_c0 + co0 + (lane & 1) * 16) = *(const v4i*)(stage + (lane >> 1) * DSP + (lane & 1) * 16);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- round 5: the PIXEL-STATIONARY form for launches of many rounds of waves (a batch of frames) ---------------------------------------
// On this part a wave's VALU instructions do not hide behind f32 MFMAs -- its own or its SIMD-mate's (the encode kernel's
// stamps) -- so what the 64 x 64 form above loses is what it ISSUES beside them: five VALU per int8 -> fp32 conversion, every pixel
// converted once per 64 columns (80 per 32 MFMAs = 15 % of a step), and the per-lane address arithmetic of ten loads per step.  Here a
// wave keeps the fp32 B operands of its 32 pixels in K / 2 registers (the encode wave form's idea: codebook_encode_wave.hip) and streams
// the weights of NP pairs of 32-column tiles past them: a pixel is converted once per 128 NP columns, the weight loads are buffer loads
// with a SCALAR running offset (one 8-byte load per lane feeds two MFMAs; the same [Cin/4][cols][k0,k2,k1,k3] array as above), the two
// tiles of a pair are two independent MFMA chains.  Same ascending-ci fma chain per output, same epilogue: bit-identical results.
constexpr int PS_NPF = 4;                     // weight quads in flight per tile

#ifdef QV2X_DPS_FINE                          // dev build (tools/dps_fine.py): s_memtime stamps of every 36th item of every Cin
__device__ long long g_dps_fine[3][1024][20];                  // (up to eight pairs per item: 2 + 2 x 8 stamps)
#define DFINE(k) do { if (lane == 0 && item % 36 == 0 && item < 36 * 1024) g_dps_fine[K == 256 ? 0 : K == 128 ? 1 : 2][item / 36][(k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DFINE(k) do { } while (0)
#endif

template <int K, int MT, bool F32IN = false>
__device__ __forceinline__ void deconv_ps_item(const DeconvArgs& a, const int item) {
    const int NP = a.np;
    const int lane = threadIdx.x & 63, par = lane >> 5, l31 = lane & 31, half = par;
    const int tiles_m = (a.M + 32 * MT - 1) / (32 * MT);
    // item -> (column chunk, pixel tile): the four waves of a workgroup stream the same weights
    const int chunk = item / tiles_m, tm = item - chunk * tiles_m;
    DFINE(0);
    // the weights as a buffer: lane (column l31, k parity par) reads the float2 (k_par, k_par + 2) of a quad of input channels
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.cin * a.ncols * 4, 0x00020000);
    const int loff = l31 * 16 + par * 8;
    const int qstride = a.ncols * 16;                                   // bytes between quads
    int wo = chunk * NP * 1024;                                         // bytes to the current pair's first column (64 columns x 16 B)
    struct W2 { float x, y; };
    auto wload = [&](int quad, int t) __attribute__((always_inline)) {  // (past the array's end: the buffer returns zeros -- the last pair's look-ahead)
        return __builtin_bit_cast(W2, __builtin_amdgcn_raw_buffer_load_b64(wrs, loff, wo + quad * qstride + t * 512, 0));
    };
    W2 ring[PS_NPF][2];
#pragma unroll
    for (int i = 0; i < PS_NPF; ++i) { ring[i][0] = wload(i, 0); ring[i][1] = wload(i, 1); }

    // this lane's MT pixels (one per 32-pixel tile), their parity's K / 2 channels as fp32: m[i][j] = channel 2 j + par
    float m[MT][K / 2];
    int pixbase[MT];            // padded output pixel index at sub-position (0, 0); -1 past the end
    if constexpr (F32IN) {
        // the input is an fp32 map [M][K] (a fused pyramid level, the decoded feature): the lane's row, sixteen float4 at a time; half-wave
        // `par` keeps k = 2 j + par
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m_raw = (tm * MT + i) * 32 + l31;
            const int mm = m_raw < a.M ? m_raw : a.M - 1;
            const int img = mm / (a.h * a.wd), rem = mm - img * (a.h * a.wd);
            const int y = rem / a.wd, x = rem - y * a.wd;
            pixbase[i] = m_raw < a.M ? (img * (a.h * a.s + 2) + y * a.s + 1) * (a.wd * a.s + 2) + x * a.s + 1 : -1;
            const v4f* src = (const v4f*)((const float*)a.in + (size_t)mm * K);
#pragma unroll
            for (int c0 = 0; c0 < K / 4; c0 += 16) {
                v4f rawf[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) rawf[c] = src[c0 + c];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float e0 = rawf[c][0], e1 = rawf[c][1], e2 = rawf[c][2], e3 = rawf[c][3];
                    m[i][2 * (c0 + c)] = par ? e1 : e0;
                    m[i][2 * (c0 + c) + 1] = par ? e3 : e2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        v4i raw[MT][K / 16];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m_raw = (tm * MT + i) * 32 + l31;
            const int mm = m_raw < a.M ? m_raw : a.M - 1;
            const int img = mm / (a.h * a.wd), rem = mm - img * (a.h * a.wd);
            const int y = rem / a.wd, x = rem - y * a.wd;
            pixbase[i] = m_raw < a.M ? (img * (a.h * a.s + 2) + y * a.s + 1) * (a.wd * a.s + 2) + x * a.s + 1 : -1;
            const v4i* src = (const v4i*)(a.in + ((size_t)(img * (a.h + 2) + y + 1) * (a.wd + 2) + x + 1) * a.cin);
#if defined(QV2X_DPS_ABL) && QV2X_DPS_ABL == 2      // dev ablation (timing only): no pixel loads
#pragma unroll
            for (int c = 0; c < K / 16; ++c) raw[i][c] = v4i{lane, c, lane + c, 7};
#else
#pragma unroll
            for (int c = 0; c < K / 16; ++c) raw[i][c] = src[c];
#endif
        }
        const float fax = (float)(a.ax - 128);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int c = 0; c < K / 16; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // the stored byte is code - 128: code = byte ^ 0x80 as an unsigned byte; code + (ax - 128) = code - zx, exact in fp32
                    const unsigned u = ((unsigned)raw[i][c][d] ^ 0x80808080u) >> (8 * par);
                    m[i][8 * c + 2 * d] = ((float)(u & 0xffu) + fax) * a.dx;
                    m[i][8 * c + 2 * d + 1] = ((float)((u >> 16) & 0xffu) + fax) * a.dx;
                }
    }
    __builtin_amdgcn_sched_barrier(0);
    DFINE(1);

    const int orow = a.wd * a.s + 2;
    const float rd = 1.0f / a.out_delta, lowc = a.relu ? a.out_zp + 8388608.0f : 8388608.0f;
    // where the next 32-column tile goes: output channels co0 .. co0 + 31 of sub-position (di, dj) -- Cout is a multiple of 32; one division
    // per item, then stepped tile by tile (wave-uniform)
    int co0, di, dj;
    {
        const int col = chunk * NP * 64, ij = col / a.cout;
        co0 = col - ij * a.cout; di = ij / a.s; dj = ij - di * a.s;
    }
    constexpr int Q = K / 4;
#pragma unroll 1
    for (int P = 0; P < NP; ++P) {
        int tco[2], tsub[2];                                            // the pair's two tiles: first channel, sub-position offset in output pixels
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            tco[t] = co0; tsub[t] = di * orow + dj;
            co0 += 32;
            if (co0 == a.cout) { co0 = 0; if (++dj == a.s) { dj = 0; ++di; } }
        }
        v4f bn[2][4];                                                   // their bias, in the accumulator's layout
        v16f acc[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;
#pragma unroll
        for (int g = 0; g < Q; ++g) {
            const W2 A0 = ring[g % PS_NPF][0], A1 = ring[g % PS_NPF][1];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, m[i][2 * g], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, m[i][2 * g], acc[i][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                          // (hipcc otherwise groups the MFMAs of one accumulator: a dependent chain)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, m[i][2 * g + 1], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, m[i][2 * g + 1], acc[i][1], 0, 0, 0);
            }
            // the bias is requested BEFORE the first look-ahead load of the next pair: loads return in order, so the epilogue's wait for it
            // leaves the whole ring in flight
            if (g == Q - PS_NPF - 1) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bn[t][q] = *(const v4f*)(a.bias + tco[t] + 8 * q + 4 * half);
            }
            // the slot is free: quad g + NPF of this pair, or of the next one (64 columns = 1024 bytes on)
#if defined(QV2X_DPS_ABL) && QV2X_DPS_ABL == 3      // dev ablation: the ring is never refilled
            if (false) {} else if (false)
#endif
            if (g + PS_NPF < Q) { ring[g % PS_NPF][0] = wload(g + PS_NPF, 0); ring[g % PS_NPF][1] = wload(g + PS_NPF, 1); }
            else { ring[g % PS_NPF][0] = wload(g + PS_NPF - Q, 2); ring[g % PS_NPF][1] = wload(g + PS_NPF - Q, 3); }
            __builtin_amdgcn_sched_barrier(0);                          // (and sinks every load to its first use)
        }
        DFINE(2 + 2 * P);
#if defined(QV2X_DPS_ABL) && QV2X_DPS_ABL == 1      // dev ablation: no epilogue (the sums stay live)
        if (acc[0][0][0] + acc[MT - 1][1][5] == 1.2345f && pixbase[0] >= 0) *(float*)(a.out + (size_t)pixbase[0] * a.out_ctotal) = acc[0][0][3] + acc[MT - 1][1][7] + bn[0][0][0];
        wo += 1024;
        continue;
#endif
        // the pair's tiles: + bias, ReLU + output quantizer; v_permlane32_swap turns the lane's four channel quads (0-3 | 8-11 | 16-19 |
        // 24-27 in the lower half-wave, +4 in the upper) into 16 consecutive channels -- one 16-byte store per lane and tile, no LDS stage
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float yv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[e] = acc[i][t][4 * q + e] + bn[t][q][e];
                    pk[q] = q_pack4(yv[0], yv[1], yv[2], yv[3], a.out_delta, rd, a.out_zp, lowc);
                }
                const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
                const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
                v4i ob;
                ob[0] = s02[0]; ob[1] = s02[1]; ob[2] = s13[0]; ob[3] = s13[1];
                if (pixbase[i] >= 0) *(v4i*)(a.out + (size_t)(pixbase[i] + tsub[t]) * a.out_ctotal + a.out_c0 + tco[t] + half * 16) = ob;
            }
        DFINE(3 + 2 * P);
        wo += 1024;
    }
}

template <int NT, bool F32IN = false>
__global__ __launch_bounds__(256) void deconv_f32_kernel(const DeconvArgs a) {
    __shared__ __attribute__((aligned(16))) int8_t stagebuf[4][32 * DSP];     // per wave: [32 pixels][32 channels] of one tile
    deconv_tile<NT, F32IN>(a, __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), stagebuf);
}

// Several deblocks in one launch (they only feed the concat, so all of them can run once the last block is done): one pool
// of wave tiles instead of three launches with three tails.
constexpr int MAX_BATCH = 4;
struct DeconvBatch {
    DeconvArgs a[MAX_BATCH];
    int tile_end[MAX_BATCH];                  // running total of wave tiles
    int n;
};

template <int NT, int MT>
__global__ __launch_bounds__(256) void deconv_f32_batch_kernel(const DeconvBatch b) {
    __shared__ __attribute__((aligned(16))) int8_t stagebuf[4][32 * DSP];
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    int l = 0, begin = 0;
    while (l < b.n - 1 && tile >= b.tile_end[l]) { begin = b.tile_end[l]; ++l; }
    if (tile >= b.tile_end[b.n - 1]) return;
    switch (l) {                              // constant indices: the argument structs stay in SGPRs / kernarg loads
        case 0: deconv_tile<NT, false, MT>(b.a[0], tile - begin, stagebuf); break;
        case 1: deconv_tile<NT, false, MT>(b.a[1], tile - begin, stagebuf); break;
        case 2: deconv_tile<NT, false, MT>(b.a[2], tile - begin, stagebuf); break;
        default: deconv_tile<NT, false, MT>(b.a[3], tile - begin, stagebuf); break;
    }
}

// the batch as items of the pixel-stationary form: (MT x 32 pixels) x (NP pairs of 32-column tiles); the host orders the layers by
// falling Cin, so the long items run first and the launch's tail is made of short ones.  Cin 64: two pixel tiles per item (64 operand
// registers; a weight load feeds four MFMAs), all 128 columns of a 64 -> 128 s = 1 deblock; Cin 128 | 256: 512 columns per item.
constexpr int ps_mt(int k) { return k == 64 ? 2 : 1; }
template <bool F32IN>
__global__ __launch_bounds__(256, 2) void deconv_ps_batch_kernel(const DeconvBatch b) {
    const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (item >= b.tile_end[b.n - 1]) return;
    int l = 0, begin = 0;
    while (l < b.n - 1 && item >= b.tile_end[l]) { begin = b.tile_end[l]; ++l; }
    const DeconvArgs& a = b.a[l];                                      // (wave-uniform index into the kernel argument: scalar loads)
    if (!F32IN && a.cin == 256) deconv_ps_item<256, ps_mt(256), false>(a, item - begin);      // (fp32 rows of 256 channels keep the wave-tile kernel: registers)
    else if (a.cin == 128) deconv_ps_item<128, ps_mt(128), F32IN>(a, item - begin);
    else deconv_ps_item<64, ps_mt(64), F32IN>(a, item - begin);
}

// does the pixel-stationary form take this layer?  Cin of 64 | 128 | 256 (the operand registers are sized at compile time)
static bool ps_takes(const DeconvArgs& a) { return a.cin == 64 || a.cin == 128 || a.cin == 256; }
// items of the batch with at most `cap` column pairs per item (a layer's np: the largest power of two <= cap dividing its pairs)
static int ps_plan(DeconvBatch& b, int n, int cap) {
    int total = 0;
    for (int i = 0; i < n; ++i) {
        DeconvArgs& a = b.a[i];
        const int pairs = a.ncols / 64, mt = ps_mt(a.cin);
        a.np = 1;
        while (a.np * 2 <= cap && pairs % (a.np * 2) == 0) a.np *= 2;
        total += ((a.M + 32 * mt - 1) / (32 * mt)) * (pairs / a.np);
        b.tile_end[i] = total;
    }
    return total;
}

static int deconv_args(const qv2x_deconv_desc* d, const int8_t* in, const float* w, const float* bias, int8_t* out, const char* who, DeconvArgs& a) {
    if (!d || !in || !w || !bias || !out) return fail(QV2X_EINVAL, "%s: null pointer", who);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->s < 1 || d->s > 8) return fail(QV2X_EINVAL, "%s: bad shape", who);
    if (d->cin % 16 || (d->s * d->s * d->cout) % 64 || d->cout % 32) return fail(QV2X_EALIGN, "%s: cin %% 16, cout %% 32, s*s*cout %% 64", who);
    if (((uintptr_t)in & 15) || ((uintptr_t)w & 15)) return fail(QV2X_EALIGN, "%s: in / w must be 16-byte aligned", who);
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "%s: out_delta must be positive", who);
    if (d->out_h != d->h * d->s || d->out_w != d->w * d->s)
        return fail(QV2X_EINVAL, "%s: destination is %d x %d, this layer writes %d x %d", who, d->out_h, d->out_w, d->h * d->s, d->w * d->s);
    if (d->out_c0 < 0 || d->out_ctotal < d->out_c0 + d->cout) return fail(QV2X_EINVAL, "%s: out channel window", who);
    if (d->out_ctotal % 16 || d->out_c0 % 16 || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15))
        return fail(QV2X_EALIGN, "%s: out_ctotal, out_c0 %% 16; out / bias 16-byte aligned (16-byte stores)", who);
    a.in = in; a.w = w; a.bias = bias; a.out = out;
    a.n = d->n; a.h = d->h; a.wd = d->w; a.cin = d->cin; a.cout = d->cout; a.s = d->s; a.ax = 128 - d->in_zx;
    a.ncols = d->s * d->s * d->cout; a.M = d->n * d->h * d->w; a.relu = d->relu;
    a.out_ctotal = d->out_ctotal; a.out_c0 = d->out_c0; a.dx = d->in_delta; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    return QV2X_OK;
}

}  // namespace qv2x

extern "C" int qv2x_deconv_i8(const qv2x_deconv_desc* d, const int8_t* in, const float* w, const float* bias, int8_t* out, void* stream) {
    using namespace qv2x;
    DeconvArgs a;
    if (int rc = deconv_args(d, in, w, bias, out, "qv2x_deconv_i8", a)) return rc;
    // 32 x 32 wave tiles: 4416 tiles of 3.9 us on 1024 SIMDs at the 25 x 88 level balance better than 2208 tiles of 7.8 us
    // (13.7 / 21.4 / 36.4 us against 15.4 / 23.7 / 39.5 us for the three deblocks)
#ifdef QV2X_DEV_KNOBS                                                  // dev build only: column tiles per wave
    static const char* ntenv = getenv("QV2X_DECONV_NT");
    const int nt = ntenv ? atoi(ntenv) : 1;
#else
    const int nt = 1;
#endif
    const int tiles = ((a.M + 31) / 32) * (a.ncols / (32 * nt));
    if (nt == 1) deconv_f32_kernel<1><<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
    else deconv_f32_kernel<2><<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_deconv_i8 launch");
}

extern "C" int qv2x_deconv_i8_batch(const qv2x_deconv_desc* descs, int n, const int8_t* const* ins, const float* const* ws,
                                    const float* const* biases, int8_t* const* outs, void* stream) {
    using namespace qv2x;
    if (!descs || !ins || !ws || !biases || !outs) return fail(QV2X_EINVAL, "qv2x_deconv_i8_batch: null pointer");
    if (n < 1 || n > MAX_BATCH) return fail(QV2X_EINVAL, "qv2x_deconv_i8_batch: 1..%d layers", MAX_BATCH);
    DeconvBatch b{};
    int total1 = 0;
    for (int i = 0; i < n; ++i) {
        if (int rc = deconv_args(&descs[i], ins[i], ws[i], biases[i], outs[i], "qv2x_deconv_i8_batch", b.a[i])) return rc;
        total1 += ((b.a[i].M + 31) / 32) * (b.a[i].ncols / 32);
    }
    // Wave tiles of 32 pixels x 32 columns balance best while a launch is a few rounds of waves (one frame: 13.7 / 21.4 / 36.4 us against
    // 15.4 / 23.7 / 39.5 with 64 columns); from ~16 tiles per SIMD on, 64 columns per wave halve the pixel loads and the int8 -> fp32
    // conversions per MFMA (every s*s*cout is a multiple of 64), and two pixel tiles per wave halve the weight loads (128 columns: 1262 us)
    // (round 3, batch of 32 frames, us for the three deblocks: 1 x 1 tiles per wave 1325, 1 x 2 columns 1209, 2 pixels x 1 1280, 2 x 2 1186)
    const int nt = total1 >= 16384 ? 2 : 1, mt = nt;
    // Round 5: the pixel-stationary items when every layer of the batch has a shape they take.  Column pairs per item: 8 (a pixel is
    // converted once per 512 columns) while that leaves 1.25 rounds of waves, else 4, 2, 1 (us per batch of 1 / 2 / 4 / 8 / 32 frames:
    // the tiles above 56 / 99 / 175 / 320 / 1155; items of 2 pairs 45 / 86 / - / - / 1077, of 4 pairs 56 / 86 / 154 / 287 / 1024,
    // of 8 pairs - / 107 / 157 / 272 / 980)
    bool ps = true;
    for (int i = 0; i < n; ++i) ps = ps && ps_takes(b.a[i]);
#ifdef QV2X_DEV_KNOBS
    static const char* psenv = getenv("QV2X_DECONV_PS");
    if (psenv && atoi(psenv) == 0) ps = false;
#endif
    if (ps) {
        for (int i = 1; i < n; ++i)                                     // falling Cin (n <= 4: insertion sort)
            for (int j = i; j > 0 && b.a[j].cin > b.a[j - 1].cin; --j) { const DeconvArgs t = b.a[j]; b.a[j] = b.a[j - 1]; b.a[j - 1] = t; }
        int cap = 8, total = ps_plan(b, n, cap);
#ifdef QV2X_DEV_KNOBS
        static const char* npenv = getenv("QV2X_DECONV_NP");
        if (npenv) { cap = atoi(npenv); total = ps_plan(b, n, cap); } else
#endif
        while (cap > 1 && total < 2560) total = ps_plan(b, n, cap >>= 1);
        b.n = n;
        deconv_ps_batch_kernel<false><<<dim3((total + 3) / 4), 256, 0, (hipStream_t)stream>>>(b);
        return hip_check(hipGetLastError(), "qv2x_deconv_i8_batch launch");
    }
    int total = 0;
    for (int i = 0; i < n; ++i) {
        total += ((b.a[i].M + 32 * mt - 1) / (32 * mt)) * (b.a[i].ncols / (32 * nt));
        b.tile_end[i] = total;
    }
    b.n = n;
    const dim3 grid((total + 3) / 4);
    if (nt == 2) deconv_f32_batch_kernel<2, 2><<<grid, 256, 0, (hipStream_t)stream>>>(b);
    else deconv_f32_batch_kernel<1, 1><<<grid, 256, 0, (hipStream_t)stream>>>(b);
    return hip_check(hipGetLastError(), "qv2x_deconv_i8_batch launch");
}

extern "C" int qv2x_deconv_f32in(const qv2x_deconv_desc* d, const float* in, const float* w, const float* bias, int8_t* out, void* stream) {
    using namespace qv2x;
    DeconvBatch b{};
    DeconvArgs& a = b.a[0];
    if (int rc = deconv_args(d, (const int8_t*)in, w, bias, out, "qv2x_deconv_f32in", a)) return rc;
    if (ps_takes(a) && a.cin <= 128) {      // round 5: the pixel-stationary items (a lane's row of the fp32 map straight into operand registers)
        int cap = 8, total = ps_plan(b, 1, cap);
        while (cap > 1 && total < 2560) total = ps_plan(b, 1, cap >>= 1);
        b.n = 1;
        deconv_ps_batch_kernel<true><<<dim3((total + 3) / 4), 256, 0, (hipStream_t)stream>>>(b);
        return hip_check(hipGetLastError(), "qv2x_deconv_f32in launch");
    }
    const int tiles = ((a.M + 31) / 32) * (a.ncols / 32);
    deconv_f32_kernel<1, true><<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_deconv_f32in launch");
}

#ifdef QV2X_DPS_FINE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_dps_fine(long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(qv2x::g_dps_fine), sizeof(long long) * 3 * 1024 * 20) == hipSuccess ? 0 : -1;
}
#endif
