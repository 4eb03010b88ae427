// f3: the grouped 3x3 convolution of QuantBottleneck (ResNeXt 32 x 4d: 32 groups of 4, 8 or 16 channels; resblock.py:69-128 under
// quant_block.py:100-131), stride 1 or 2, + folded BN bias + ReLU + output quantizer.
//
// 36 .. 144 MACs per output: no GEMM to speak of -- the layer is HBM/L2-bound byte work, so it runs on the VALU's packed int8 dot
// product (v_dot4_i32_i8), not on the matrix cores.  One thread = one output pixel x 16 consecutive channels (4, 2 or 1 groups): it
// reads its nine 16-byte window pieces from the padded i8 BEV, the weights of the chunk are wave-uniform (blockIdx.y) and come
// through scalar loads; T = sum (x - zx)(w - zw) exactly via the same gemmlowp identity as the MFMA kernels
// (sum x_s w_s + aw * sum x_s + corr), four outputs requantize at a time (q_pack4) and leave as one 16-byte store.
#include "common.h"

namespace qv2x {
namespace {

struct GArgs {
    const int8_t* in; const int* w; const float* scale; const int32_t* corr; const int32_t* aw; const float* bias; int8_t* out;
    int n, h, wd, c, stride, ho, wo, M, relu;
    float out_delta, out_zp;
};

template <int CG>
__global__ __launch_bounds__(256) void gconv3x3_i8_kernel(const GArgs a) {
    constexpr int DW = CG / 4;                                       // dwords of one group's channels
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const int chunk = blockIdx.y;
    const int img = m / (a.ho * a.wo), rem = m - img * (a.ho * a.wo);
    const int yo = rem / a.wo, xo = rem - yo * a.wo;
    const int8_t* base = a.in + ((size_t)(img * (a.h + 2) + yo * a.stride) * (a.wd + 2) + xo * a.stride) * a.c + chunk * 16;
    v4i win[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) win[t] = *(const v4i*)(base + ((size_t)(t / 3) * (a.wd + 2) + t % 3) * a.c);
    const int* __restrict__ wc = a.w + (size_t)chunk * (16 * 9 * DW);
    const int c0 = chunk * 16;
    const float rd = 1.0f / a.out_delta, lo = a.relu ? 0.0f : -3.0e38f;
    v4i outw;
#pragma unroll
    for (int gi = 0; gi < 16 / CG; ++gi) {
        int sum = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int dw = 0; dw < DW; ++dw) sum = __builtin_amdgcn_sdot4(win[t][gi * DW + dw], 0x01010101, sum, false);
#pragma unroll
        for (int o4 = 0; o4 < CG / 4; ++o4) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = gi * CG + o4 * 4 + e;                // channel inside the chunk
                int acc = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int dw = 0; dw < DW; ++dw) acc = __builtin_amdgcn_sdot4(win[t][gi * DW + dw], wc[(col * 9 + t) * DW + dw], acc, false);
                const int T = acc + a.aw[c0 + col] * sum + a.corr[c0 + col];
                y[e] = fmaxf(a.bias[c0 + col] + (float)T * a.scale[c0 + col], lo);
            }
            outw[(gi * CG + o4 * 4) >> 2] = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp);
        }
    }
    *(v4i*)(a.out + ((size_t)(img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1) * a.c + c0) = outw;
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_gconv3x3_i8(const qv2x_gconv_desc* d, const int8_t* in, const int8_t* w_chunk, const float* scale, const int32_t* corr,
                                const int32_t* aw, const float* bias, int8_t* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w_chunk || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_gconv3x3_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->stride != 1 && d->stride != 2)) return fail(QV2X_EINVAL, "qv2x_gconv3x3_i8: bad shape / stride");
    if (d->c % 16 || (d->cg != 4 && d->cg != 8 && d->cg != 16)) return fail(QV2X_EALIGN, "qv2x_gconv3x3_i8: channels %% 16, 4 | 8 | 16 channels per group");
    if (((uintptr_t)in & 15) || ((uintptr_t)w_chunk & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_gconv3x3_i8: 16-byte aligned pointers");
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_gconv3x3_i8: out_delta must be positive");
    GArgs a;
    a.in = in; a.w = (const int*)w_chunk; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    a.n = d->n; a.h = d->h; a.wd = d->w; a.c = d->c; a.stride = d->stride;
    a.ho = (d->h - 1) / d->stride + 1; a.wo = (d->w - 1) / d->stride + 1; a.M = d->n * a.ho * a.wo;
    a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    dim3 grid((a.M + 255) / 256, a.c / 16);
    hipStream_t st = (hipStream_t)stream;
    if (d->cg == 4) gconv3x3_i8_kernel<4><<<grid, 256, 0, st>>>(a);
    else if (d->cg == 8) gconv3x3_i8_kernel<8><<<grid, 256, 0, st>>>(a);
    else gconv3x3_i8_kernel<16><<<grid, 256, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_gconv3x3_i8 launch");
}
