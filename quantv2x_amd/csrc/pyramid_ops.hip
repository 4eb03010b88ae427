// f3: the small HBM-bound steps of the HEAL Pyramid path around the convolutions.
//   qv2x_codebook_decode_f32   UMGMQuantizer.decode (codebook.py:339-343) as table look-ups, any width D (64 on this model):
//                              out[r] = ((bias + T0[c0]) + T1[c1]) + T2[c2] in that order
//   qv2x_occ_score_i8          the 1x1 occupancy head of one pyramid level (QuantModule, quant_block.py:475-479) on codes + its
//                              output quantizer + score = sigmoid(occ) + 1e-4 (:509) through a 256-entry table
#include "common.h"

namespace qv2x {
namespace {

struct DecArgs {
    const uint8_t* codes; const float* lut; const float* bias; float* out;
    int64_t agent_stride, level_stride;
    int rows, hw, levels, kc, d;
};

__global__ __launch_bounds__(256) void decode_lut_kernel(const DecArgs a) {
    const int per = a.d >> 2;                                      // float4 pieces per row
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)a.rows * per) return;
    const int r = (int)(id / per), p = (int)(id - (int64_t)r * per);
    const int ag = r / a.hw, cell = r - ag * a.hw;
    v4f v = *(const v4f*)(a.bias + 4 * p);
    for (int l = 0; l < a.levels; ++l) {
        const int c = a.codes[ag * a.agent_stride + l * a.level_stride + cell];
        const v4f t = *(const v4f*)(a.lut + ((size_t)(l * a.kc + c) * a.d) + 4 * p);
        v = v + t;
    }
    *(v4f*)(a.out + (size_t)r * a.d + 4 * p) = v;
}

struct OccArgs {
    const int8_t* in; const int8_t* w; const float* lut; float* score; uint8_t* code;
    int n, h, wd, c, M, aw, corr;
    float scale, bias, out_delta, out_zp;
};

__global__ __launch_bounds__(256) void occ_score_kernel(const OccArgs a) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const int img = m / (a.h * a.wd), rem = m - img * (a.h * a.wd);
    const int y = rem / a.wd, x = rem - y * a.wd;
    const int8_t* src = a.in + ((size_t)(img * (a.h + 2) + y + 1) * (a.wd + 2) + x + 1) * a.c;
    int acc = 0, sum = 0;
    for (int k = 0; k < a.c; k += 16) {
        const v4i px = *(const v4i*)(src + k);
        const v4i wv = *(const v4i*)(a.w + k);                      // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_sdot4(px[q], wv[q], acc, false);
            sum = __builtin_amdgcn_sdot4(px[q], 0x01010101, sum, false);
        }
    }
    const int T = acc + a.aw * sum + a.corr;
    const float yv = a.bias + (float)T * a.scale;
    const int code = (int)q_code_mul(yv, a.out_delta, a.out_zp);
    a.score[m] = a.lut[code];
    if (a.code) a.code[m] = (uint8_t)code;
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_codebook_decode_f32(const uint8_t* codes, int64_t agent_stride, int64_t level_stride, int agents, int hw, int levels,
                                        int kc, int d, const float* lut, const float* bias, float* out, void* stream) {
    using namespace qv2x;
    if (!codes || !lut || !bias || !out) return fail(QV2X_EINVAL, "qv2x_codebook_decode_f32: null pointer");
    if (agents <= 0 || hw <= 0 || levels < 1 || levels > 16 || kc < 1 || kc > 256 || d <= 0 || d % 4)
        return fail(QV2X_EINVAL, "qv2x_codebook_decode_f32: bad sizes (agents %d, hw %d, levels %d, kc %d, d %d)", agents, hw, levels, kc, d);
    if (((uintptr_t)lut & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_codebook_decode_f32: 16-byte aligned tables / output");
    DecArgs a{codes, lut, bias, out, agent_stride, level_stride, agents * hw, hw, levels, kc, d};
    const int64_t work = (int64_t)a.rows * (d / 4);
    decode_lut_kernel<<<(unsigned)((work + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_decode_f32 launch");
}

extern "C" int qv2x_occ_score_i8(const qv2x_occ_desc* d, const int8_t* in, const int8_t* w, const float* score_lut, float* score,
                                 uint8_t* occ_code, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w || !score_lut || !score) return fail(QV2X_EINVAL, "qv2x_occ_score_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->c % 16) return fail(QV2X_EINVAL, "qv2x_occ_score_i8: bad shape (channels %% 16)");
    if (((uintptr_t)in & 15) || ((uintptr_t)w & 15)) return fail(QV2X_EALIGN, "qv2x_occ_score_i8: 16-byte aligned map / weights");
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_occ_score_i8: out_delta must be positive");
    OccArgs a{in, w, score_lut, score, occ_code, d->n, d->h, d->w, d->c, d->n * d->h * d->w, d->aw, d->corr, d->scale, d->bias, d->out_delta, d->out_zp};
    occ_score_kernel<<<(a.M + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_occ_score_i8 launch");
}
