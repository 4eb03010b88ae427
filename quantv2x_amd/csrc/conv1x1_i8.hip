// f3: the 1x1 convolutions of the HEAL Pyramid model under QuantModel -- conv1 / conv3 and the strided shortcut of QuantBottleneck
// and QuantBasicBlock (opencood/quant/quant_block.py:68-131 over quant_layer.py:391-410) -- on v_mfma_i32_32x32x32_i8.
//
// GEMM view: rows = output channels (the WEIGHTS are the A operand), columns = output pixels, K = Cin <= 512.  One wave = 32 pixels
// x 64 channels: a lane reads its pixel's 16 K-bytes per MFMA step straight from the padded i8 BEV (the map is read once per 64
// output channels, from L2 after the first), the weights arrive pre-packed in fragment order (16 B per lane, coalesced, L2
// resident: <= 128 KB per layer).  The lane then owns ONE pixel and 16 channels per tile in four runs of four, so the epilogue is
// per-lane: the zero-point correction uses the lane's own window sum, four results requantize at a time (q_pack4) and the end of a
// residual block -- out = quant(relu(conv3(x) + shortcut)) -- is fused: the shortcut comes in as fp32 (the strided 1x1 branch, or the
// decoded feature in front of the first block) or as the block input's codes, dequantized in the epilogue.
#include "common.h"

namespace qv2x {
namespace {

struct C1Args {
    const int8_t* in; const int8_t* w; const float* scale; const int32_t* corr; const int32_t* aw; const float* bias;
    const void* res; void* out;
    int n, h, wd, cin, cout, stride, ho, wo, M, relu, out_ctotal, out_c0, res_ax;
    float out_delta, out_zp, res_delta;
};

template <int MODE>
__global__ __launch_bounds__(256) void conv1x1_i8_kernel(const C1Args a) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (tile * 32 >= a.M) return;
    const int cb = blockIdx.y;                                     // 64 output channels
    const int m = tile * 32 + l31;
    const bool valid = m < a.M;
    const int mm = valid ? m : a.M - 1;
    const int img = mm / (a.ho * a.wo), rem = mm - img * (a.ho * a.wo);
    const int yo = rem / a.wo, xo = rem - yo * a.wo;
    const int8_t* src = a.in + ((size_t)(img * (a.h + 2) + yo * a.stride + 1) * (a.wd + 2) + xo * a.stride + 1) * a.cin + 16 * half;
    const int ksteps = a.cin >> 5;
    const int8_t* wp = a.w + ((size_t)(cb * 2) * ksteps) * 1024 + lane * 16;

    v16i acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    int xs = 0;

    for (int ks = 0; ks < ksteps; ++ks) {
        const v4i fb = *(const v4i*)(src + 32 * ks);
        const v4i fa0 = *(const v4i*)(wp + (size_t)ks * 1024);
        const v4i fa1 = *(const v4i*)(wp + (size_t)(ksteps + ks) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(fb[q], 0x01010101, xs, false);
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb, acc[1], 0, 0, 0);
    }
    const int tot = xs + __shfl_xor(xs, 32);                       // sum of the pixel's (code - 128) over Cin
    if (!valid) return;

    const float rd = 1.0f / a.out_delta, lo = a.relu ? 0.0f : -3.0e38f;
    const size_t opix = (size_t)(img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1;       // padded output pixel
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = cb * 64 + j * 32 + 8 * g + 4 * half;
            const v4i xa = *(const v4i*)(a.aw + ch), xc = *(const v4i*)(a.corr + ch);
            const v4f xsc = *(const v4f*)(a.scale + ch), xb = *(const v4f*)(a.bias + ch);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int T = acc[j][4 * g + e] + xa[e] * tot + xc[e];
                y[e] = xb[e] + (float)T * xsc[e];
            }
            if (MODE == 1) {                                       // disable_act_quant: the fp32 map [M][Cout]
                v4f o = {y[0], y[1], y[2], y[3]};
                *(v4f*)((float*)a.out + (size_t)m * a.cout + ch) = o;
                continue;
            }
            if (MODE == 2) {
                const v4f r = *(const v4f*)((const float*)a.res + (size_t)m * a.cout + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = y[e] + r[e];
            }
            if (MODE == 3) {
                const int rw = *(const int*)((const int8_t*)a.res + opix * a.cout + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = y[e] + (float)(((rw << (24 - 8 * e)) >> 24) + a.res_ax) * a.res_delta;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], lo);
            *(int*)((int8_t*)a.out + opix * a.out_ctotal + a.out_c0 + ch) = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp);
        }
    }
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_conv1x1_i8(const qv2x_conv1x1_desc* d, const int8_t* in, const int8_t* w_frag, const float* scale, const int32_t* corr,
                               const int32_t* aw, const float* bias, const void* res, void* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w_frag || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->stride != 1 && d->stride != 2)) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: bad shape / stride");
    if (d->cin % 32 || d->cin > 1024 || d->cout % 64) return fail(QV2X_EALIGN, "qv2x_conv1x1_i8: cin %% 32 (<= 1024), cout %% 64");
    if (d->mode < 0 || d->mode > 3) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: mode 0..3");
    if ((d->mode >= 2) && !res) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: mode %d needs the shortcut", d->mode);
    if (((uintptr_t)in & 15) || ((uintptr_t)w_frag & 15) || ((uintptr_t)out & 15) || ((uintptr_t)res & 15) || ((uintptr_t)scale & 15) ||
        ((uintptr_t)corr & 15) || ((uintptr_t)aw & 15) || ((uintptr_t)bias & 15)) return fail(QV2X_EALIGN, "qv2x_conv1x1_i8: 16-byte aligned pointers");
    C1Args a;
    a.in = in; a.w = w_frag; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.res = res; a.out = out;
    a.n = d->n; a.h = d->h; a.wd = d->w; a.cin = d->cin; a.cout = d->cout; a.stride = d->stride;
    a.ho = (d->h - 1) / d->stride + 1; a.wo = (d->w - 1) / d->stride + 1; a.M = d->n * a.ho * a.wo;
    a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp; a.res_ax = 128 - d->res_zx; a.res_delta = d->res_delta;
    a.out_ctotal = d->mode == 1 ? d->cout : d->out_ctotal; a.out_c0 = d->mode == 1 ? 0 : d->out_c0;
    if (d->mode != 1) {
        if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: out_delta must be positive");
        if (d->out_ctotal % 4 || d->out_c0 % 4 || d->out_ctotal < d->out_c0 + d->cout) return fail(QV2X_EALIGN, "qv2x_conv1x1_i8: out channel window");
    }
    dim3 grid(((a.M + 31) / 32 + 3) / 4, a.cout / 64);
    hipStream_t st = (hipStream_t)stream;
    switch (d->mode) {
        case 0: conv1x1_i8_kernel<0><<<grid, 256, 0, st>>>(a); break;
        case 1: conv1x1_i8_kernel<1><<<grid, 256, 0, st>>>(a); break;
        case 2: conv1x1_i8_kernel<2><<<grid, 256, 0, st>>>(a); break;
        default: conv1x1_i8_kernel<3><<<grid, 256, 0, st>>>(a); break;
    }
    return hip_check(hipGetLastError(), "qv2x_conv1x1_i8 launch");
}
