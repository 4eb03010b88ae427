// a6: UMGMQuantizer.encode (codebook.py:330-337 -> :231-239 -> :106-131), m = 1, D = 256, fused over all
// residual levels on v_mfma_f32_32x32x2_f32.
//
// Per level:  z = stage(x);  q = qhead(z);  d_k = (|q|^2 + |C_k|^2) - 2 (q . C_k);  code = first argmin_k;
//             x <- lhead(z) - C[code]          (reference op order; every dot product is an ascending-k fp32
//             fma chain with acc0 = bias, which is what the f32 MFMA evaluates -- bit-exact against
//             oracle/qv2x_oracle.c:orc_codebook_encode).
// A workgroup owns 64 BEV cells: x / z / q live in two LDS buffers (row stride 260 floats: ds_read_b128 of a
// 16-lane group lands on 16 distinct 16-byte slots), the 256 x 256 weight matrices stream from L2 as
// [K/4][col][4] float4 (coalesced per wave), register double buffered.  Wave w owns output columns
// [64w, 64w + 64) for all 64 rows (2 x 2 MFMA tiles); for the distance GEMM codes [32w, 32w + 32).
// The argmin runs on the accumulator registers: half-wave butterfly on (distance, index) with ties to the lower
// index, then a 4-way combine across waves through LDS.
#include "common.h"

namespace qv2x {

constexpr int ER = 64;              // rows per workgroup
constexpr int LDF = 260;            // LDS row stride in floats
constexpr int D = 256;

struct EncArgs {
    const int8_t* in; uint8_t* codes;
    const float* lvl[4];
    int n, h, w, levels, kc, ax, M;
    float dx;
};

__device__ __forceinline__ size_t off_stage_w() { return 0; }
__device__ __host__ __forceinline__ int64_t level_floats(int kc) { return 3LL * (D * D + D) + (int64_t)D * kc + (int64_t)kc * D + kc; }

// out[64 rows][cols 64*wave .. +64) = in[64][256] . W^T, acc0 = bias.  W packed [64][256][4].
__device__ __forceinline__ void gemm_64x64(const float* __restrict__ src, const float4* __restrict__ wp, const float* __restrict__ bias,
                                           int wave, int lane, v16f (&acc)[2][2]) {
    const int par = lane >> 5, col = wave * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float b = bias[col + j * 32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
    }
    float4 bn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bn[j] = wp[col + j * 32];
    for (int q = 0; q < 64; ++q) {
        float4 bc[2] = {bn[0], bn[1]};
        if (q + 1 < 64) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bn[j] = wp[(size_t)(q + 1) * D + col + j * 32];
        }
        float4 av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *(const float4*)(src + (i * 32 + (lane & 31)) * LDF + q * 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float a = half ? (par ? av[i].w : av[i].z) : (par ? av[i].y : av[i].x);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float b = half ? (par ? bc[j].w : bc[j].z) : (par ? bc[j].y : bc[j].x);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, int wave, int lane, const v16f (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dst[(i * 32 + mfma32_row(r, lane)) * LDF + wave * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
}

__global__ __launch_bounds__(256) void codebook_encode_kernel(const EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;                       // x, then q, then next x
    float* bufB = smem + ER * LDF;            // z
    float* x2 = smem + 2 * ER * LDF;          // [64]
    float* pval = x2 + ER;                    // [4][64]
    int* pidx = (int*)(pval + 4 * ER);        // [4][64]
    int* code_s = pidx + 4 * ER;              // [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * ER;

    {   // ---- load 64 rows x 256 channels from the i8 BEV, dequantize ---------------------------------
        const int row = tid >> 2, part = tid & 3;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int img = m / (a.h * a.w), rem = m - img * (a.h * a.w);
        const int y = rem / a.w, x = rem - y * a.w;
        const int8_t* src = a.in + ((size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1) * D + part * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4i raw = *(const v4i*)(src + c * 16);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xs = (raw[e >> 2] << (24 - (e & 3) * 8)) >> 24;
                bufA[row * LDF + part * 64 + c * 16 + e] = (float)(xs + a.ax) * a.dx;
            }
        }
    }
    __syncthreads();

    v16f acc[2][2];
    for (int l = 0; l < a.levels; ++l) {
        const float* W = a.lvl[l];
        const float* stage_w = W;
        const float* stage_b = stage_w + D * D;
        const float* qhead_w = stage_b + D;
        const float* qhead_b = qhead_w + D * D;
        const float* lhead_w = qhead_b + D;
        const float* lhead_b = lhead_w + D * D;
        const float* cbp = lhead_b + D;                       // [64][kc][4]
        const float* cb = cbp + (size_t)D * a.kc;             // [kc][256]
        const float* c2 = cb + (size_t)a.kc * D;              // [kc]

        gemm_64x64(bufA, (const float4*)stage_w, stage_b, wave, lane, acc);      // z = stage(x)
        store_tile(bufB, wave, lane, acc);
        __syncthreads();
        gemm_64x64(bufB, (const float4*)qhead_w, qhead_b, wave, lane, acc);      // q = qhead(z)
        __syncthreads();                                                          // all reads of x (bufA) done in GEMM 1; safe to overwrite
        store_tile(bufA, wave, lane, acc);
        __syncthreads();

        {   // |q|^2: four 64-wide ascending fma chains per row, combined (s0 + s1) + (s2 + s3)
            const int row = tid >> 2, part = tid & 3;
            const float* qr = bufA + row * LDF + part * 64;
            float s = 0.0f;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) s = fmaf(qr[i], qr[i], s);
            const float s01 = s + __shfl_xor(s, 1);
            const float tot = s01 + __shfl_xor(s01, 2);
            if (part == 0) x2[row] = tot;
        }
        __syncthreads();

        // ---- distances for codes [32*wave, +32) and the per-wave argmin ---------------------------------
        if (wave * 32 < a.kc) {
            const int par = lane >> 5, code = wave * 32 + (lane & 31);
            v16f dacc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[i][r] = 0.0f;
            const float4* cp = (const float4*)cbp;
            float4 bn = cp[code];
            for (int q = 0; q < 64; ++q) {
                const float4 bc = bn;
                if (q + 1 < 64) bn = cp[(size_t)(q + 1) * a.kc + code];
                float4 av[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = *(const float4*)(bufA + (i * 32 + (lane & 31)) * LDF + q * 4);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const float b = half ? (par ? bc.w : bc.z) : (par ? bc.y : bc.x);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float av_ = half ? (par ? av[i].w : av[i].z) : (par ? av[i].y : av[i].x);
                        dacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, b, dacc[i], 0, 0, 0);
                    }
                }
            }
            const float c2v = c2[code];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + mfma32_row(r, lane);
                    float dv = (x2[row] + c2v) - 2.0f * dacc[i][r];
                    int di = code;
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {      // butterfly inside each 32-lane half
                        const float ov = __shfl_xor(dv, off);
                        const int oi = __shfl_xor(di, off);
                        const bool take = (ov < dv) || (ov == dv && oi < di);
                        dv = take ? ov : dv;
                        di = take ? oi : di;
                    }
                    if ((lane & 31) == 0) { pval[wave * ER + row] = dv; pidx[wave * ER + row] = di; }
                }
        }
        __syncthreads();
        if (tid < ER) {
            float bv = pval[tid]; int bi = pidx[tid];
            for (int wv = 1; wv * 32 < a.kc; ++wv) {
                const float ov = pval[wv * ER + tid]; const int oi = pidx[wv * ER + tid];
                if (ov < bv) { bv = ov; bi = oi; }                 // strict: earlier wave = lower indices wins ties
            }
            code_s[tid] = bi;
            if (m0 + tid < a.M) a.codes[(size_t)l * a.M + m0 + tid] = (uint8_t)bi;
        }
        __syncthreads();

        if (l + 1 < a.levels) {      // x <- lhead(z) - C[code]
            gemm_64x64(bufB, (const float4*)lhead_w, lhead_b, wave, lane, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + mfma32_row(r, lane), col = wave * 64 + j * 32 + (lane & 31);
                        bufA[row * LDF + col] = acc[i][j][r] - cb[(size_t)code_s[row] * D + col];
                    }
            __syncthreads();
        }
    }
}

__global__ void codebook_c2_kernel(const float* __restrict__ cb, int kc, float* __restrict__ c2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= kc) return;
    float s[4];
    for (int q = 0; q < 4; ++q) {
        float acc = 0.0f;
        for (int i = 0; i < 64; ++i) { const float v = cb[(size_t)k * D + q * 64 + i]; acc = fmaf(v, v, acc); }
        s[q] = acc;
    }
    c2[k] = (s[0] + s[1]) + (s[2] + s[3]);
}

}  // namespace qv2x

extern "C" int qv2x_codebook_c2_f32(const float* codebook, int kc, float* c2, void* stream) {
    using namespace qv2x;
    if (!codebook || !c2 || kc <= 0) return fail(QV2X_EINVAL, "qv2x_codebook_c2_f32: bad arguments");
    codebook_c2_kernel<<<(kc + 63) / 64, 64, 0, (hipStream_t)stream>>>(codebook, kc, c2);
    return hip_check(hipGetLastError(), "qv2x_codebook_c2_f32 launch");
}

extern "C" int64_t qv2x_codebook_level_floats(int kc) { return qv2x::level_floats(kc); }

extern "C" int qv2x_codebook_encode_f32(const qv2x_encode_desc* d, const int8_t* in, const float* const* level_weights,
                                        uint8_t* codes, void* stream) {
    using namespace qv2x;
    if (!d || !in || !level_weights || !codes) return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 4) return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: bad shape");
    if (d->kc < 32 || d->kc > 128 || d->kc % 32) return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: dict_size must be 32, 64, 96 or 128 (got %d)", d->kc);
    if ((uintptr_t)in & 15) return fail(QV2X_EALIGN, "qv2x_codebook_encode_f32: in must be 16-byte aligned");
    EncArgs a;
    a.in = in; a.codes = codes; a.n = d->n; a.h = d->h; a.w = d->w; a.levels = d->levels; a.kc = d->kc;
    a.ax = 128 - d->in_zx; a.dx = d->in_delta; a.M = d->n * d->h * d->w;
    for (int l = 0; l < 4; ++l) {
        a.lvl[l] = l < d->levels ? level_weights[l] : nullptr;
        if (l < d->levels && (!a.lvl[l] || ((uintptr_t)a.lvl[l] & 15))) return fail(QV2X_EALIGN, "qv2x_codebook_encode_f32: level %d weights null or unaligned", l);
    }
    const size_t smem = (size_t)(2 * ER * LDF + ER + 4 * ER + 4 * ER + ER) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        int rc = hip_check(hipFuncSetAttribute((const void*)codebook_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                           "qv2x_codebook_encode_f32 smem attribute");
        if (rc) return rc;
        attr_set = true;
    }
    codebook_encode_kernel<<<(a.M + ER - 1) / ER, 256, smem, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_f32 launch");
}
