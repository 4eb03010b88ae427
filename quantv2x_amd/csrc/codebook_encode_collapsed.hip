// OPT-IN, NOT the parity configuration: UMGMQuantizer.encode (opencood/models/sub_modules/codebook.py:330-337 -> :231-239 -> :106-131)
// with its affine heads collapsed on the host.  Every head of the encoder is affine and the residual step subtracts a codeword, so
//     distance_l[k] - |q_l|^2  =  G_l[k] . x  +  g_l[k]  +  sum_{j < l} T_lj[code_j][k]
// where x is the SHARED feature the first level sees, G_l / g_l fold stage, quantization head, codebook and the latent heads of the levels
// in front, and T_lj is what choosing code_j at level j changes at level l.  |q_l|^2 does not depend on k and drops out of the argmin.
// One 256 -> levels * Kc GEMM per cell (6.3x fewer flops than the reference's eleven chained 256-wide GEMMs) followed by a short
// argmin chain over the tables.  fp32 re-association means the argmin can differ from the reference's where the two best distances are
// closer than rounding error: qv2x_codebook_encode_f32 stays the shipped and tested path; this entry exists to MEASURE that difference
// (tools/bench_collapsed_encode.py) and as an explicit opt-in (`engine.encode_mode = "collapsed"`).
//
// Kernel: persistent workgroups of levels * Kc / 32 waves (12 at 3 x 128).  Wave w keeps the 256 x 32 slice of G for score columns
// [32 w, 32 w + 32) in REGISTERS (128 floats per lane, the B operand of v_mfma_f32_32x32x2_f32) for the whole launch: no weight traffic
// at all.  Per 32-cell tile: the cells' 256 uint8 codes sit in LDS as bytes (8 KB; the dequantization x = delta (code - zp) is folded into
// G and g), every wave runs 128 MFMAs with A converted from the bytes on the fly, scores land in LDS, then waves 0-7 run the argmin chain
// (four cells per wave, two candidates per lane, lowest index wins ties; it adds the table rows of the codes already chosen) while waves
// 8-11 bring in the next tile's bytes.  Always 12 waves: with fewer score columns the spare ones hold no slice of G.
#include "common.h"

namespace qv2x {
namespace {

struct CollArgs {
    const int8_t* in; const float* gpack; const float* bias; const float* tables; uint8_t* codes;
    int n, h, w, hw, R, levels, kc, nct;
};

constexpr int NR = 96;                                               // values of G a lane keeps in registers (of 128)
constexpr int RP = 260;                                              // byte pitch of a cell's 256 codes in LDS (65 dwords: 32 rows on 32 banks)

__device__ __forceinline__ void wave_argmin(float& v, int& idx) {     // lexicographic (value, index) minimum over the wave
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

__global__ __launch_bounds__(768) void encode_collapsed_kernel(const CollArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int SP = a.nct * 32 + 1;                                   // score pitch in floats
    uint8_t* rows = smem;                                            // [32][RP]
    float* scores = (float*)(smem + 32 * RP);                    // [32][SP]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, par = lane >> 5;
    const int tiles = (a.R + 31) >> 5;
    if ((int)blockIdx.x >= tiles) return;

    // G does not fit the register file next to everything else (98 304 floats against 4 x 512 registers x 64 lanes per CU, three waves per
    // SIMD): a lane keeps the first NR of its 128 values in registers and the last 128 - NR in LDS (8 KB per wave, read once per tile)
    const bool gemm = wave < a.nct;
    float* wl = scores + 32 * SP + wave * (128 - NR) * 64;           // [128 - NR][64]
    float bw[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) bw[i] = gemm ? a.gpack[((size_t)wave * 128 + i) * 64 + lane] : 0.0f;
    if (gemm) {
#pragma unroll
        for (int i = NR; i < 128; ++i) wl[(i - NR) * 64 + lane] = a.gpack[((size_t)wave * 128 + i) * 64 + lane];
    }
    const float b0 = gemm ? a.bias[wave * 32 + r] : 0.0f;

    // cell m -> its 256 bytes in the padded i8 map, 16-byte pieces; the 256 threads of waves 8-11 move two pieces each
    auto load_tile = [&](int tile) {
        const int t = threadIdx.x - 512;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int id = t + 256 * e, lr = id >> 4, lp = id & 15;
            const int m = min(tile * 32 + lr, a.R - 1);
            const int img = m / a.hw, rem = m - img * a.hw, y = rem / a.w, x = rem - y * a.w;
            const v4i c = *(const v4i*)(a.in + ((size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1) * 256 + lp * 16);
            int* dst = (int*)(rows + lr * RP + lp * 16);
            dst[0] = c[0] ^ (int)0x80808080; dst[1] = c[1] ^ (int)0x80808080; dst[2] = c[2] ^ (int)0x80808080; dst[3] = c[3] ^ (int)0x80808080;    // (code - 128) -> code
        }
    };
    if (wave >= 8) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();                                             // the tile's bytes are in LDS; the previous tile's argmin is done with `scores`
        if (gemm) {
            v16f acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;              // (the bias is added at the end: an all-zero C operand costs no registers)
            const uint8_t* ar = rows + r * RP;
            auto rd = [&](int q4) -> v4i {
                v4i d = {*(const int*)(ar + q4 * 16), *(const int*)(ar + q4 * 16 + 4), *(const int*)(ar + q4 * 16 + 8), *(const int*)(ar + q4 * 16 + 12)};
                return d;
            };
            v4i dn = rd(0);
#pragma unroll
            for (int q4 = 0; q4 < 16; ++q4) {                        // 16 bytes = 8 MFMA steps; the next 16 are read while these multiply
                const v4i d = dn;
                if (q4 + 1 < 16) dn = rd(q4 + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned u = (unsigned)d[j];
                    const float a0 = (float)((u >> (par ? 8 : 0)) & 255u), a1 = (float)((u >> (par ? 24 : 16)) & 255u);
                    const int i0 = q4 * 8 + 2 * j;
                    const float w0 = i0 < NR ? bw[i0 < NR ? i0 : 0] : wl[(i0 - NR) * 64 + lane];
                    const float w1 = i0 + 1 < NR ? bw[i0 + 1 < NR ? i0 + 1 : 0] : wl[(i0 + 1 - NR) * 64 + lane];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w1, acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) scores[mfma32_row(e, lane) * SP + wave * 32 + r] = acc[e] + b0;
        }
        __syncthreads();                                             // all scores of the tile are in LDS, every wave is done with its bytes
        if (wave >= 8 && tile + (int)gridDim.x < tiles) load_tile(tile + gridDim.x);
        if (wave < 8) {                                              // four cells per wave
            int code[4][3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                if (l < a.levels) {
                    float v[4];
                    int ix[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* sr = scores + (wave * 4 + j) * SP + l * a.kc;
                        float v0 = lane < a.kc ? sr[lane] : INFINITY, v1 = lane + 64 < a.kc ? sr[lane + 64] : INFINITY;
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            if (p < l) {
                                const float* t = a.tables + ((size_t)(l * (l - 1) / 2 + p) * a.kc + code[j][p]) * a.kc;
                                if (lane < a.kc) v0 = v0 + t[lane];
                                if (lane + 64 < a.kc) v1 = v1 + t[lane + 64];
                            }
                        }
                        v[j] = v0; ix[j] = lane;
                        if (v1 < v0) { v[j] = v1; ix[j] = lane + 64; }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wave_argmin(v[j], ix[j]);
                        code[j][l] = __builtin_amdgcn_readfirstlane(ix[j]);
                    }
                }
            }
            if (lane < 4 * a.levels) {                               // lane -> (cell j, level l)
                const int j = lane & 3, l = lane >> 2;
                const int m = tile * 32 + wave * 4 + j;
                int c = 0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int ll = 0; ll < 3; ++ll)
                        if (jj == j && ll == l) c = code[jj][ll];
                if (m < a.R) a.codes[(size_t)l * a.R + m] = (uint8_t)c;
            }
        }
    }
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_codebook_encode_collapsed_f32(const qv2x_encode_desc* d, const int8_t* in, const float* g_packed, const float* bias,
                                                  const float* tables, uint8_t* codes, void* stream) {
    using namespace qv2x;
    if (!d || !in || !g_packed || !bias || !codes) return fail(QV2X_EINVAL, "qv2x_codebook_encode_collapsed_f32: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 3) return fail(QV2X_EINVAL, "qv2x_codebook_encode_collapsed_f32: 1..3 levels");
    if (d->segs > 1) return fail(QV2X_EINVAL, "qv2x_codebook_encode_collapsed_f32: seg_num 1 only (the opt-in collapsed form; the exact entry takes seg_num 1 | 2 | 4)");
    if (d->kc < 32 || d->kc > 128 || d->kc % 32) return fail(QV2X_EINVAL, "qv2x_codebook_encode_collapsed_f32: dict_size must be 32, 64, 96 or 128 (got %d)", d->kc);
    if (d->levels > 1 && !tables) return fail(QV2X_EINVAL, "qv2x_codebook_encode_collapsed_f32: the residual levels need their tables");
    if (((uintptr_t)in & 15) || ((uintptr_t)g_packed & 15)) return fail(QV2X_EALIGN, "qv2x_codebook_encode_collapsed_f32: 16-byte aligned pointers");
    CollArgs a;
    a.in = in; a.gpack = g_packed; a.bias = bias; a.tables = tables; a.codes = codes;
    a.n = d->n; a.h = d->h; a.w = d->w; a.hw = d->h * d->w; a.R = d->n * a.hw; a.levels = d->levels; a.kc = d->kc; a.nct = d->levels * d->kc / 32;
    const int lds = 32 * RP + 32 * (a.nct * 32 + 1) * 4 + 12 * (128 - NR) * 64 * 4;
    if (int rc = hip_check(hipFuncSetAttribute((const void*)encode_collapsed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "collapsed encode LDS size")) return rc;
    int dev = 0, cus = 256, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    const int tiles = (a.R + 31) / 32;
    encode_collapsed_kernel<<<min(tiles, cus), 768, lds, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_collapsed_f32 launch");
}
