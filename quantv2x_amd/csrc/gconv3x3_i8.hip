// f3: the grouped 3x3 convolution of QuantBottleneck (ResNeXt 32 x 4d: 32 groups of 4, 8 or 16 channels; resblock.py:69-128 under
// quant_block.py:100-131), stride 1 or 2, + folded BN bias + ReLU + output quantizer, on v_mfma_i32_32x32x32_i8.
//
// 36 .. 144 MACs per output is no GEMM to speak of, but the VALU's packed dot product (v_dot4_i32_i8) issues at half rate on gfx950
// (tools/probes/dot4_probe.hip) and a first version built on it ran at 75 us for 36 MB of activations.  A 32-channel slab of the layer
// IS a dense 32 -> 32 3x3 convolution whose weight matrix is block diagonal (8, 4 or 2 groups): nine MFMAs (one per tap, K = the slab's
// 32 input channels) per 32 pixels, the off-diagonal blocks zero.  1/8 .. 1/2 of the multiplies are structural zeros; the matrix
// core still finishes the layer an order of magnitude sooner than the VALU, and the layer goes back to being the HBM-bound byte
// work it is.  The window sums of the gemmlowp identity (per pixel AND group) come from nine more MFMAs against the block-diagonal
// matrix of ones -- exact, and already laid out per output channel.
//
// A workgroup = a 4 x 32 patch of output pixels x FOUR slabs = 128 channels, one slab per wave: the input patch
// ((3 S + 3) x (31 S + 3) pixels x 128 bytes, 29 / 84 KB) is staged in LDS once -- every window piece would cross L2 nine times
// otherwise, and with one slab per workgroup every 128-byte line came up from L2 for 32 useful bytes (54 us for a 36 MB map) -- a wave
// walks the patch's rows as 32-pixel MFMA tiles with its slab's weights (9 taps as A fragments, 36 VGPRs) in registers.  The WEIGHTS
// are the A operand: a lane then owns one pixel and 16 channels in four runs of four (q_pack4).
#include "common.h"

namespace qv2x {
namespace {

struct GArgs {
    const int8_t* in; const int8_t* w; const float* scale; const int32_t* corr; const int32_t* aw; const float* bias; int8_t* out;
    int n, h, wd, c, cg, stride, ho, wo, M, relu;
    float out_delta, out_zp;
};

template <int S>
__global__ __launch_bounds__(256) void gconv3x3_i8_kernel(const GArgs a) {
    constexpr int TH = 4, TW = 32, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    constexpr int SP = 48;                                           // output staging pitch per pixel: 32 channel bytes + 16 (bank spread)
    constexpr int PP = 128 + 16;                                     // patch pitch per pixel: 128 channel bytes + 16 (the 32 lanes of a fragment read land on distinct bank groups)
    __shared__ __attribute__((aligned(16))) int8_t patch[PH * PW * PP + 4 * 32 * SP + 4 * 128 * 4];
    int* cst = (int*)(patch + PH * PW * PP + 4 * 32 * SP);            // [aw | corr | scale | bias][128] of this workgroup's channels
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int cbase = blockIdx.y * 128, nslab = min(4, (a.c - cbase) / 32);
    const int slab = blockIdx.y * 4 + wave, c0 = slab * 32;
    const int tiles_x = (a.wo + TW - 1) / TW, tiles_y = (a.ho + TH - 1) / TH;
    const int img = blockIdx.x / (tiles_x * tiles_y), trem = blockIdx.x - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    {   // input patch: padded rows y0 S .. + PH, columns x0 S .. + PW (clamped into the padded map at the ragged edges: unused there)
        const int8_t* ibase = a.in + (size_t)img * (a.h + 2) * (a.wd + 2) * a.c + cbase;
        const int pieces = nslab * 2;                                // 16-byte pieces per pixel
        constexpr int ITER = (PH * PW * 8 + 255) / 256;
        v4i tmp[ITER];                                               // every request in flight before the first LDS write
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = it * 256 + threadIdx.x, pix = min(i >> 3, PH * PW - 1), piece = i & 7;
            const int py = pix / PW, px = pix - py * PW;
            const int yy = min(y0 * S + py, a.h + 1), xx = min(x0 * S + px, a.wd + 1);
            tmp[it] = *(const v4i*)(ibase + ((size_t)yy * (a.wd + 2) + xx) * a.c + (piece < pieces ? piece : 0) * 16);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = it * 256 + threadIdx.x, pix = i >> 3, piece = i & 7;
            if (pix < PH * PW && piece < pieces) *(v4i*)(patch + pix * PP + piece * 16) = tmp[it];
        }
    }
    const bool active = wave < nslab;                                // (a layer whose channel count is not a multiple of 128)
    // the slab's weights: [tap][lane][16 B], lane = 32 * ((ci / 16) & 1) + co % 32
    v4i wf[9];
    const int8_t* wp = a.w + (size_t)(active ? slab : 0) * (9 * 1024) + lane * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[t] = *(const v4i*)(wp + t * 1024);
    // block-diagonal ones: byte b of this lane's fragment is input channel 16 half + b against output channel l31
    v4i ones;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int wv = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) wv |= (((16 * half + 4 * q + e) / a.cg) == (l31 / a.cg) ? 1 : 0) << (8 * e);
        ones[q] = wv;
    }
    {   // epilogue constants of the workgroup's (up to) 128 channels
        const int which = threadIdx.x >> 6, t = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = cbase + t + 64 * k;
            const int cc = c < a.c ? c : 0;
            cst[which * 128 + t + 64 * k] = which == 0 ? a.aw[cc] : which == 1 ? a.corr[cc] : which == 2 ? __float_as_int(a.scale[cc]) : __float_as_int(a.bias[cc]);
        }
    }
    int8_t* stage = patch + PH * PW * PP + wave * (32 * SP);
    __syncthreads();
    if (!active) return;
    const float rd = 1.0f / a.out_delta, lo = a.relu ? 0.0f : -3.0e38f;
    for (int ty = 0; ty < TH; ++ty) {
        if (y0 + ty >= a.ho) break;
        v16i acc, sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0; sum[r] = 0; }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const v4i fb = *(const v4i*)(patch + ((ty * S + t / 3) * PW + l31 * S + t % 3) * PP + wave * 32 + half * 16);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[t], fb, acc, 0, 0, 0);
            sum = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, fb, sum, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wave * 32 + 8 * g + 4 * half;
            const v4i xa = *(const v4i*)(cst + cl), xc = *(const v4i*)(cst + 128 + cl);
            const v4f xsc = *(const v4f*)(cst + 256 + cl), xb = *(const v4f*)(cst + 384 + cl);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int T = acc[4 * g + e] + xa[e] * sum[4 * g + e] + xc[e];
                y[e] = fmaxf(xb[e] + (float)T * xsc[e], lo);
            }
            *(int*)(stage + l31 * SP + 8 * g + 4 * half) = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // the row's 32 pixels x 32 bytes leave as 16-byte stores: pixel lane >> 1, piece lane & 1
        const int yo = y0 + ty, xo = x0 + (lane >> 1);
        if (yo < a.ho && xo < a.wo)
            *(v4i*)(a.out + ((size_t)(img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1) * a.c + c0 + (lane & 1) * 16) =
                *(const v4i*)(stage + (lane >> 1) * SP + (lane & 1) * 16);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_gconv3x3_i8(const qv2x_gconv_desc* d, const int8_t* in, const int8_t* w_frag, const float* scale, const int32_t* corr,
                                const int32_t* aw, const float* bias, int8_t* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w_frag || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_gconv3x3_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->stride != 1 && d->stride != 2)) return fail(QV2X_EINVAL, "qv2x_gconv3x3_i8: bad shape / stride");
    if (d->c % 32 || (d->cg != 4 && d->cg != 8 && d->cg != 16)) return fail(QV2X_EALIGN, "qv2x_gconv3x3_i8: channels %% 32, 4 | 8 | 16 channels per group");
    if (((uintptr_t)in & 15) || ((uintptr_t)w_frag & 15) || ((uintptr_t)out & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)corr & 15) ||
        ((uintptr_t)aw & 15) || ((uintptr_t)bias & 15)) return fail(QV2X_EALIGN, "qv2x_gconv3x3_i8: 16-byte aligned pointers");
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_gconv3x3_i8: out_delta must be positive");
    GArgs a;
    a.in = in; a.w = w_frag; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    a.n = d->n; a.h = d->h; a.wd = d->w; a.c = d->c; a.cg = d->cg; a.stride = d->stride;
    a.ho = (d->h - 1) / d->stride + 1; a.wo = (d->w - 1) / d->stride + 1; a.M = d->n * a.ho * a.wo;
    a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    const int th = 4;
    dim3 grid(a.n * ((a.ho + th - 1) / th) * ((a.wo + 31) / 32), (a.c + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    if (a.stride == 1) gconv3x3_i8_kernel<1><<<grid, 256, 0, st>>>(a);
    else gconv3x3_i8_kernel<2><<<grid, 256, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_gconv3x3_i8 launch");
}
