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
    DeconvArgs a;
    if (int rc = deconv_args(d, (const int8_t*)in, w, bias, out, "qv2x_deconv_f32in", a)) return rc;
    const int tiles = ((a.M + 31) / 32) * (a.ncols / 32);
    deconv_f32_kernel<1, true><<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_deconv_f32in launch");
}
