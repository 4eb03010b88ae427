// a11: 1x1 detection heads (cls | reg | dir stacked on the channel axis) on fp32 rows, fake-quant weights,
// per-channel output quantizer.  f32 MFMA fma chains (acc0 = bias), results transposed through LDS so that the
// NCHW store is 128 B contiguous per channel.  Also: decode-only LUT kernel for the *_single heads.
#include "common.h"

namespace qv2x {

// wave = (32-row tile, 32-column tile): with <= 96 stacked output channels that is up to three waves per row tile, which
// keeps all 1024 SIMDs of the chip busy at 35 200 rows (a 96-column wave tile left the second round almost empty).
__global__ __launch_bounds__(256) void heads_f32_kernel(const float* __restrict__ x, int R, int hw, int cout, int cout_pad,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ da, const float* __restrict__ za,
                                                        float* __restrict__ out) {
    __shared__ float tr[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nct = cout_pad >> 5;
    int tile = blockIdx.x * 4 + wave;
    tile = __builtin_amdgcn_readfirstlane(tile);
    const int tm = tile / nct, ct = tile - tm * nct;
    if (tm * 32 >= R) return;
    const int par = lane >> 5;
    int m = tm * 32 + (lane & 31);
    m = m < R ? m : R - 1;
    const float4* xr = (const float4*)(x + (size_t)m * 256);

    v16f acc;
    {
        const float b = bias[ct * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
    }
    // weights: [64][cout_pad][k0, k2, k1, k3] -> one float2 per lane per k-quad; activations: the lane's row, float4 per
    // k-quad (both half-waves read the same 16 B and keep their parity's pair).  Eight k-quads are requested ahead of
    // each MFMA block (register double buffer pinned with sched_barrier).
    const float2* wl = (const float2*)w + (size_t)(ct * 32 + (lane & 31)) * 2 + par;
    auto loadA = [&](float4 (&dst)[8], int q0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) dst[t] = xr[q0 + t];
    };
    auto loadB = [&](float2 (&dst)[8], int q0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) dst[t] = wl[(size_t)(q0 + t) * cout_pad * 2];
    };
    auto block = [&](const float4 (&av)[8], const float2 (&bv)[8]) {
        float a0[8], a1[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { a0[t] = par ? av[t].y : av[t].x; a1[t] = par ? av[t].w : av[t].z; }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], bv[t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], bv[t].y, acc, 0, 0, 0);
        }
    };
    float4 xa[8], xb[8];
    float2 wa[8], wb[8];
    loadA(xa, 0); loadB(wa, 0);
    for (int q0 = 0; q0 < 64; q0 += 16) {
        loadA(xb, q0 + 8); loadB(wb, q0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        block(xa, wa);
        __builtin_amdgcn_sched_barrier(0);
        if (q0 + 16 < 64) { loadA(xa, q0 + 16); loadB(wa, q0 + 16); }
        __builtin_amdgcn_sched_barrier(0);
        block(xb, wb);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const int co = ct * 32 + (lane & 31);
        const float d = da[co], z = za[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = acc[r];
            if (d > 0.0f) y = (q_code(y, d, z) - z) * d;
            tr[wave][mfma32_row(r, lane)][lane & 31] = y;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // lane -> cell lane&31 (128 B contiguous per channel in the NCHW output), channels (lane>>5) + 2*c
        const int mm = tm * 32 + (lane & 31);
        const int bi = mm / hw, cell = mm - bi * hw;
        float* ob = out + (size_t)bi * cout * hw + cell;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ch = ct * 32 + par + 2 * c;
            if (ch < cout && mm < R) ob[(size_t)ch * hw] = tr[wave][lane & 31][par + 2 * c];
        }
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
    if (!x || !w || !bias || !da || !za || !out) return fail(QV2X_EINVAL, "qv2x_heads_f32: null pointer");
    if (R <= 0 || hw <= 0 || R % hw || cout <= 0 || cout > cout_pad || cout_pad % 32 || cout_pad > 96)
        return fail(QV2X_EINVAL, "qv2x_heads_f32: R=%d hw=%d cout=%d cout_pad=%d (cout_pad in {32, 64, 96})", R, hw, cout, cout_pad);
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return fail(QV2X_EALIGN, "qv2x_heads_f32: x / w must be 16-byte aligned");
    const int tiles = ((R + 31) / 32) * (cout_pad / 32);
    heads_f32_kernel<<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, R, hw, cout, cout_pad, w, bias, da, za, out);
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
