// f3: the grouped 3x3 convolution of QuantBottleneck (ResNeXt 32 x 4d: 32 groups of 4, 8 or 16 channels; resblock.py:69-128 under
// quant_block.py:100-131), stride 1 or 2, + folded BN bias + ReLU + output quantizer.
//
// 36 .. 144 MACs per output: no GEMM to speak of -- the layer is HBM/L2-bound byte work, so it runs on the VALU's packed int8 dot
// product (v_dot4_i32_i8), not on the matrix cores.  One thread = one output pixel x 16 consecutive channels (4, 2 or 1 groups);
// T = sum (x - zx)(w - zw) exactly via the same gemmlowp identity as the MFMA kernels
// (sum x_s w_s + aw * sum x_s + corr), four outputs requantize at a time (q_pack4) and leave as one 16-byte store.
#include "common.h"

namespace qv2x {
namespace {

struct GArgs {
    const int8_t* in; const int* w; const float* scale; const int32_t* corr; const int32_t* aw; const float* bias; int8_t* out;
    int n, h, wd, c, stride, ho, wo, M, relu;
    float out_delta, out_zp;
};

// A workgroup = an 8 x 32 patch of output pixels x one 16-channel chunk.  Its input patch ((8 S + 2) x (32 S + 2) pixels x 16 bytes) is
// staged in LDS once -- read straight from the map every window piece crossed L2 nine times (20 MB for a 2 MB map at 25 x 88) -- with
// the chunk's weights (16 outputs x 9 taps x CG bytes) and epilogue constants (as scalar loads they cost one s_waitcnt round trip per
// batch).  Thread = one output pixel: nine ds_read_b128 for the window, weights as LDS broadcasts.
template <int CG, int S>
__global__ __launch_bounds__(256) void gconv3x3_i8_kernel(const GArgs a) {
    constexpr int DW = CG / 4;                                       // dwords of one group's channels
    constexpr int WN = 16 * 9 * DW;                                  // weight dwords of the chunk
    constexpr int TH = 8, TW = 32, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    __shared__ __attribute__((aligned(16))) int wsm[WN + 64 + PH * PW * 4];
    v4i* patch = (v4i*)(wsm + WN + 64);
    const int chunk = blockIdx.y, c0 = chunk * 16;
    const int tiles_x = (a.wo + TW - 1) / TW, tiles_y = (a.ho + TH - 1) / TH;
    const int img = blockIdx.x / (tiles_x * tiles_y), trem = blockIdx.x - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    {
        const int* __restrict__ wc = a.w + (size_t)chunk * WN;
        for (int i = threadIdx.x; i < WN; i += 256) wsm[i] = wc[i];
        if (threadIdx.x < 16) {
            wsm[WN + threadIdx.x] = a.aw[c0 + threadIdx.x];
            wsm[WN + 16 + threadIdx.x] = a.corr[c0 + threadIdx.x];
            wsm[WN + 32 + threadIdx.x] = __float_as_int(a.scale[c0 + threadIdx.x]);
            wsm[WN + 48 + threadIdx.x] = __float_as_int(a.bias[c0 + threadIdx.x]);
        }
        // input patch: padded rows y0 S .. + PH, columns x0 S .. + PW (clamped into the padded map at the ragged edges: unused there)
        const int8_t* ibase = a.in + (size_t)img * (a.h + 2) * (a.wd + 2) * a.c + c0;
        for (int i = threadIdx.x; i < PH * PW; i += 256) {
            const int py = i / PW, px = i - py * PW;
            const int yy = min(y0 * S + py, a.h + 1), xx = min(x0 * S + px, a.wd + 1);
            patch[i] = *(const v4i*)(ibase + ((size_t)yy * (a.wd + 2) + xx) * a.c);
        }
    }
    __syncthreads();
    const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
    const int yo = y0 + ty, xo = x0 + tx;
    v4i win[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) win[t] = patch[(ty * S + t / 3) * PW + tx * S + t % 3];
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
                    for (int dw = 0; dw < DW; ++dw) acc = __builtin_amdgcn_sdot4(win[t][gi * DW + dw], wsm[(col * 9 + t) * DW + dw], acc, false);
                const int T = acc + wsm[WN + col] * sum + wsm[WN + 16 + col];
                y[e] = fmaxf(__int_as_float(wsm[WN + 48 + col]) + (float)T * __int_as_float(wsm[WN + 32 + col]), lo);
            }
            outw[(gi * CG + o4 * 4) >> 2] = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp);
        }
    }
    if (yo < a.ho && xo < a.wo) *(v4i*)(a.out + ((size_t)(img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1) * a.c + c0) = outw;
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
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(a.n * ((a.ho + 7) / 8) * ((a.wo + 31) / 32), a.c / 16);
#define QV2X_GCONV(CG) do { if (a.stride == 1) gconv3x3_i8_kernel<CG, 1><<<grid, 256, 0, st>>>(a); else gconv3x3_i8_kernel<CG, 2><<<grid, 256, 0, st>>>(a); } while (0)
    if (d->cg == 4) QV2X_GCONV(4);
    else if (d->cg == 8) QV2X_GCONV(8);
    else QV2X_GCONV(16);
#undef QV2X_GCONV
    return hip_check(hipGetLastError(), "qv2x_gconv3x3_i8 launch");
}
