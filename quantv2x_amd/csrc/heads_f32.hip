// a11: 1x1 detection heads (cls | reg | dir stacked on the channel axis) on fp32 rows, fake-quant weights,
// per-channel output quantizer.  f32 MFMA fma chains (acc0 = bias), results transposed through LDS so that the
// NCHW store is 128 B contiguous per channel.  Also: decode-only LUT kernel for the *_single heads.
#include "common.h"

namespace qv2x {

template <int NT>
__global__ __launch_bounds__(256) void heads_f32_kernel(const float* __restrict__ x, int R, int hw, int cout, int cout_pad,
                                                        const float4* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ da, const float* __restrict__ za,
                                                        float* __restrict__ out) {
    __shared__ float tr[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tm = blockIdx.x * 4 + wave;
    tm = __builtin_amdgcn_readfirstlane(tm);
    if (tm * 32 >= R) return;
    const int par = lane >> 5;
    int m = tm * 32 + (lane & 31);
    m = m < R ? m : R - 1;
    const float4* xr = (const float4*)(x + (size_t)m * 256);

    v16f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float b = bias[t * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b;
    }
#pragma unroll 4
    for (int q = 0; q < 64; ++q) {                 // k = 4q .. 4q+3
        const float4 av = xr[q];
        float4 bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = w[(size_t)q * cout_pad + t * 32 + (lane & 31)];
        const float a0 = par ? av.y : av.x, a1 = par ? av.w : av.z;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, par ? bv[t].y : bv[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, par ? bv[t].w : bv[t].z, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = t * 32 + (lane & 31);
        const float d = da[co], z = za[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = acc[t][r];
            if (d > 0.0f) y = (q_code(y, d, z) - z) * d;
            tr[wave][mfma32_row(r, lane)][lane & 31] = y;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // lane -> row (cell) lane&31, channels (lane>>5) + 2*c
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ch = t * 32 + par + 2 * c;
            const int mm = tm * 32 + (lane & 31);
            if (ch < cout && mm < R) {
                const int bi = mm / hw, cell = mm - bi * hw;
                out[((size_t)bi * cout + ch) * hw + cell] = tr[wave][lane & 31][par + 2 * c];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
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

}  // namespace qv2x

extern "C" int qv2x_heads_f32(const float* x, int R, int hw, int cout, int cout_pad, const float* w, const float* bias,
                              const float* da, const float* za, float* out, void* stream) {
    using namespace qv2x;
    if (!x || !w || !bias || !da || !za || !out) return fail(QV2X_EINVAL, "qv2x_heads_f32: null pointer");
    if (R <= 0 || hw <= 0 || R % hw || cout <= 0 || cout > cout_pad || cout_pad % 32 || cout_pad > 96)
        return fail(QV2X_EINVAL, "qv2x_heads_f32: R=%d hw=%d cout=%d cout_pad=%d (cout_pad in {32, 64, 96})", R, hw, cout, cout_pad);
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return fail(QV2X_EALIGN, "qv2x_heads_f32: x / w must be 16-byte aligned");
    const int blocks = ((R + 31) / 32 + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    const float4* w4 = (const float4*)w;
    if (cout_pad == 32) heads_f32_kernel<1><<<blocks, 256, 0, st>>>(x, R, hw, cout, cout_pad, w4, bias, da, za, out);
    else if (cout_pad == 64) heads_f32_kernel<2><<<blocks, 256, 0, st>>>(x, R, hw, cout, cout_pad, w4, bias, da, za, out);
    else heads_f32_kernel<3><<<blocks, 256, 0, st>>>(x, R, hw, cout, cout_pad, w4, bias, da, za, out);
    return hip_check(hipGetLastError(), "qv2x_heads_f32 launch");
}

extern "C" int qv2x_decode_lut_f32(const uint8_t* codes, int R, int levels, int kc, const float* lut, const float* lut_bias,
                                   float* out, void* stream) {
    using namespace qv2x;
    if (!codes || !lut || !lut_bias || !out) return fail(QV2X_EINVAL, "qv2x_decode_lut_f32: null pointer");
    if (R <= 0 || levels < 1 || levels > 4 || kc < 1 || kc > 256) return fail(QV2X_EINVAL, "qv2x_decode_lut_f32: bad sizes");
    decode_lut_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(codes, R, levels, kc, (const float4*)lut, (const float4*)lut_bias, (float4*)out);
    return hip_check(hipGetLastError(), "qv2x_decode_lut_f32 launch");
}
