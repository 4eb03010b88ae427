// f3: UMGMQuantizer.encode for the Pyramid model's 64-wide codebook (heter_pyramid_collab_codebook_mc.py:25-51, codebook.py:330-337),
// seg_num m = 1 | 2 | 4 (the extended codebook of codebook_encode.hip: m * kc rows, segment s in dims [s d, (s + 1) d)), fused over the residual levels on v_mfma_f32_32x32x2_f32 -- the D = 64 sibling of codebook_encode.hip (same op order, same
// ascending-k fma chains with acc0 = bias, bit-exact against oracle/qv2x_oracle.c:orc_codebook_encode_d).
//
// 11x fewer MACs per row than D = 256 and every matrix is 16 KB, so the shape changes: a workgroup of FOUR waves owns 64 BEV cells;
// each wave owns one 32 x 32 tile of the 64 x 64 GEMM outputs (row tile = wave >> 1, column tile = wave & 1) and two 32 x 32
// (rows x codes) tiles of the distance GEMM.  x / z / q live in two LDS buffers (row stride 66 floats), the weights stream from L2
// as [K/4][col][k0, k2, k1, k3] with the next four k-quads requested ahead of the current MFMAs; 34 KB of LDS: four workgroups per CU.
#include "common.h"

namespace qv2x {
namespace {

constexpr int D = 64, ER = 64, LDF = 66;
__device__ __forceinline__ int kpos(int c) { return (c & ~3) | ((c & 1) << 1) | ((c >> 1) & 1); }

struct Enc64Args {
    const int8_t* in; const float* in_f32; uint8_t* codes;        // in_f32 != null: the rows come as fp32 (un-quantized model)
    const float* lvl[4];
    int n, h, w, cin_total, levels, kc, ax, M;      // kc: codes per segment and level
    int segs, ke;                                   // seg_num (m); ke = segs * kc rows of the extended codebook
    float dx;
};
constexpr int MAX_TILES = 16;                       // 32-code tiles of the extended codebook: ke <= 512

// all 16 k-quads of this lane's weight column (16 float2) + its bias: requested BEFORE the barrier that precedes the GEMM, so the L2 round
// trip overlaps the previous phase's tile store and the barrier wait
struct TileW { float2 s[4][4]; float b; };
__device__ __forceinline__ TileW tile_weights(const float2* __restrict__ wp, const float* __restrict__ bias, int ctile, int lane) {
    const int par = lane >> 5, col = ctile * 32 + (lane & 31);
    TileW w;
    w.b = bias[col];
    const float2* wl = wp + (size_t)col * 2 + par;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) w.s[g][t] = wl[(size_t)(g * 4 + t) * D * 2];
    return w;
}

// this wave's 32 x 32 tile of in[64][64] . W^T (+ bias): rows [32 rt, +32), columns [32 ctile, +32)
__device__ __forceinline__ void gemm_tile(const float* __restrict__ src, const TileW& w, int rt, int lane, v16f& acc) {
    const int par = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = w.b;
    const float* al = src + (rt * 32 + (lane & 31)) * LDF + 2 * par;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float2 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = *(const float2*)(al + (g * 4 + t) * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, w.s[g][t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, w.s[g][t].y, acc, 0, 0, 0);
        }
    }
}

__device__ __forceinline__ unsigned long long dist_key(float d, int code) {
    unsigned b = __builtin_bit_cast(unsigned, d);
    b ^= (unsigned)((int)b >> 31) | 0x80000000u;
    return ((unsigned long long)b << 32) | (unsigned)code;
}

template <int CTRL>
__device__ __forceinline__ void keymin_dpp(unsigned long long& k) {
    const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    k = o < k ? o : k;
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, int rt, int ctile, int lane, const v16f& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(rt * 32 + mfma32_row(r, lane)) * LDF + kpos(ctile * 32 + (lane & 31))] = acc[r];
}

__global__ __launch_bounds__(256, 4) void codebook_encode64_kernel(const Enc64Args a) {
    __shared__ __attribute__((aligned(16))) float smem[2 * ER * LDF + 4 * ER + 2 * MAX_TILES * ER + 4 * ER];
    float* bufA = smem;                       // x, then q, then next x
    float* bufB = smem + ER * LDF;            // z
    float* x2 = smem + 2 * ER * LDF;          // [segs][ER]
    unsigned long long* pkey = (unsigned long long*)(x2 + 4 * ER);  // [MAX_TILES code tiles][ER]
    int* code_s = (int*)(pkey + MAX_TILES * ER);                    // [segs][ER]: rows of the extended codebook

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rt = wave >> 1, ctile = wave & 1;
    const int m0 = blockIdx.x * ER;
    {   // 64 rows x 64 channels from the i8 BEV, dequantized: four threads per row, 16 channels each
        const int row = tid >> 2, part = tid & 3;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int img = m / (a.h * a.w), rem = m - img * (a.h * a.w);
        const int y = rem / a.w, x = rem - y * a.w;
        const size_t pixel = (size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1;
        if (a.in_f32) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const v4f v = *(const v4f*)(a.in_f32 + pixel * a.cin_total + part * 16 + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) bufA[row * LDF + kpos(part * 16 + c * 4 + e)] = v[e];
            }
        } else {
            const v4i raw = *(const v4i*)(a.in + pixel * a.cin_total + part * 16);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xs = (raw[e >> 2] << (24 - (e & 3) * 8)) >> 24;
                bufA[row * LDF + kpos(part * 16 + e)] = (float)(xs + a.ax) * a.dx;
            }
        }
    }
    TileW tw = tile_weights((const float2*)a.lvl[0], a.lvl[0] + D * D, ctile, lane);          // stage_w / stage_b of level 0
    __syncthreads();

    v16f acc;
    for (int l = 0; l < a.levels; ++l) {
        const float* W = a.lvl[l];
        const float* stage_w = W;
        const float* stage_b = stage_w + D * D;
        const float* qhead_w = stage_b + D;
        const float* qhead_b = qhead_w + D * D;
        const float* lhead_w = qhead_b + D;
        const float* lhead_b = lhead_w + D * D;
        const float* cbp = lhead_b + D;                       // [16][ke][4]
        const float* cb = cbp + (size_t)D * a.ke;             // [ke][64]
        const float* c2 = cb + (size_t)a.ke * D;              // [ke]

        gemm_tile(bufA, tw, rt, lane, acc);                                          // z = stage(x)
        tw = tile_weights((const float2*)qhead_w, qhead_b, ctile, lane);
        store_tile(bufB, rt, ctile, lane, acc);
        __syncthreads();
        gemm_tile(bufB, tw, rt, lane, acc);                                          // q = qhead(z)
        if (l + 1 < a.levels) tw = tile_weights((const float2*)lhead_w, lhead_b, ctile, lane);   // used after the argmin
        store_tile(bufA, rt, ctile, lane, acc);
        __syncthreads();
        if (tid < ER * a.segs) {      // |q_s|^2: one ascending fma chain over the segment's 64 / segs dims per (segment, row)
            const int row = tid % ER, sg = tid / ER, nq = 16 / a.segs;
            const float2* qr = (const float2*)(bufA + row * LDF) + 2 * sg * nq;
            float s = 0.0f;
            for (int j = 0; j < nq; ++j) {
                const float2 e = qr[2 * j], o = qr[2 * j + 1];
                s = fmaf(e.x, e.x, s); s = fmaf(o.x, o.x, s); s = fmaf(e.y, e.y, s); s = fmaf(o.y, o.y, s);
            }
            x2[sg * ER + row] = s;
        }
        __syncthreads();

        // ---- distances: this wave's row tile against code tiles 2 (wave & 1), 2 (wave & 1) + 1 of every round of four tiles ------------
        for (int c = 0; c < 2 * ((a.ke + 127) >> 7); ++c) {
            const int ct = 4 * (c >> 1) + 2 * ctile + (c & 1);
            if (ct * 32 < a.ke) {
                const int par = lane >> 5, code = ct * 32 + (lane & 31);
                const float* x2r = x2 + (a.segs == 1 ? 0 : (ct * 32) / a.kc) * ER;
                v16f dacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[r] = 0.0f;
                const float2* cl = (const float2*)cbp + (size_t)code * 2 + par;
                const float* al = bufA + (rt * 32 + (lane & 31)) * LDF + 2 * par;
                float2 s[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int t = 0; t < 4; ++t) s[g][t] = cl[(size_t)(g * 4 + t) * a.ke * 2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float2 av[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[t] = *(const float2*)(al + (g * 4 + t) * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, s[g][t].x, dacc, 0, 0, 0);
                        dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, s[g][t].y, dacc, 0, 0, 0);
                    }
                }
                const float c2v = c2[code];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + mfma32_row(r, lane);
                    unsigned long long key = dist_key((x2r[row] + c2v) - 2.0f * dacc[r], code);
                    keymin_dpp<0x108>(key); keymin_dpp<0x104>(key); keymin_dpp<0x102>(key); keymin_dpp<0x101>(key);
                    const int klo = (int)(unsigned)key, khi = (int)(unsigned)(key >> 32);
                    const unsigned lo16 = (unsigned)__builtin_amdgcn_readlane(klo, 16), hi16 = (unsigned)__builtin_amdgcn_readlane(khi, 16);
                    const unsigned lo48 = (unsigned)__builtin_amdgcn_readlane(klo, 48), hi48 = (unsigned)__builtin_amdgcn_readlane(khi, 48);
                    const unsigned long long o = lane < 32 ? (((unsigned long long)hi16 << 32) | lo16) : (((unsigned long long)hi48 << 32) | lo48);
                    key = o < key ? o : key;
                    if ((lane & 31) == 0) pkey[ct * ER + row] = key;
                }
            }
        }
        __syncthreads();
        if (tid < ER * a.segs) {
            const int row = tid % ER, sg = tid / ER;
            const int tps = a.segs == 1 ? (a.ke + 31) >> 5 : a.kc >> 5;   // 32-code tiles per segment
            unsigned long long bk = pkey[(sg * tps) * ER + row];
            for (int wv = 1; wv < tps; ++wv) {
                const unsigned long long ok = pkey[(sg * tps + wv) * ER + row];
                bk = ok < bk ? ok : bk;                                // code tiles ascend: ties go to the lower code
            }
            const int bi = (int)(unsigned)bk;                          // row of the extended codebook: sg * kc + code
            code_s[sg * ER + row] = bi;
            if (m0 + row < a.M) a.codes[((size_t)l * a.segs + sg) * a.M + m0 + row] = (uint8_t)(bi - sg * a.kc);
        }
        __syncthreads();

        if (l + 1 < a.levels) {      // x <- lhead(z) - C[code]
            const int col = ctile * 32 + (lane & 31);
            float cv[16];                                                             // the chosen codewords' entries, ahead of the GEMM
            const int* cs = code_s + ((col * a.segs) >> 6) * ER;                      // the column's segment
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = cb[(size_t)cs[rt * 32 + mfma32_row(r, lane)] * D + col];
            gemm_tile(bufB, tw, rt, lane, acc);
            tw = tile_weights((const float2*)a.lvl[l + 1], a.lvl[l + 1] + D * D, ctile, lane);   // the next level's stage
#pragma unroll
            for (int r = 0; r < 16; ++r) bufA[(rt * 32 + mfma32_row(r, lane)) * LDF + kpos(col)] = acc[r] - cv[r];
            __syncthreads();
        }
    }
}

// ---- round 5: ONE WAVE per 32 cells (the D = 64 sibling of codebook_encode_wave.hip) -------------------------------------------------------
// The workgroup form above gives each of its four waves one 32 x 32 tile of every 64 x 64 GEMM -- 32 MFMAs in one dependent chain between
// two workgroup barriers, the operands re-read from LDS: 0.26 of the f32 peak, 125-170 us per frame of the Pyramid model whatever its size
// below one round.  Here the product is transposed (weights = A operand, straight from the SAME [K/4][col][k0, k2, k1, k3] arrays by 8-byte
// loads; the 32 cells' activations = B operand in 32 registers), both 32-channel tiles of a GEMM run as two independent chains, the C -> B
// transposition is one trip through the wave's own 8.5 KB of LDS, |q|^2 one chain per (segment, cell) on that row, the argmin a running
// strict < inside the lane.  Same ascending-k fma chains with acc0 = bias, same distance expression: the planes are the workgroup form's.
constexpr int WRS = 68;                                                // LDS floats per cell row

template <bool F32IN>
__global__ __launch_bounds__(64, 2) void codebook_encode64_wave_kernel(const Enc64Args a) {
    __shared__ __attribute__((aligned(16))) float smem[32 * WRS];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int m0 = (int)blockIdx.x * 32;
    float* const row = smem + j * WRS;

    float xq[32], z[32];                                               // B operands: m[i] = value[cell j][2 i + h]
    {
        int m = m0 + j;
        m = m < a.M ? m : a.M - 1;
        const int img = m / (a.h * a.w), rem = m - img * (a.h * a.w);
        const int y = rem / a.w, x = rem - y * a.w;
        const size_t pixel = (size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1;
        if (F32IN) {
            const v4f* src = (const v4f*)(a.in_f32 + pixel * a.cin_total);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const v4f v = src[c];
                const float e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                xq[2 * c] = h ? e1 : e0; xq[2 * c + 1] = h ? e3 : e2;
            }
        } else {
            const v4i* src = (const v4i*)(a.in + pixel * a.cin_total);
            const float fax = (float)(a.ax - 128);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const v4i raw = src[c];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // the stored byte is code - 128: code = byte ^ 0x80 as an unsigned byte; code + (ax - 128) = xs + ax, exact in fp32
                    const int w32 = raw[d];
                    const unsigned u = ((unsigned)w32 ^ 0x80808080u) >> (8 * h);
                    xq[8 * c + 2 * d] = ((float)(u & 0xffu) + fax) * a.dx;
                    xq[8 * c + 2 * d + 1] = ((float)((u >> 16) & 0xffu) + fax) * a.dx;
                }
            }
        }
    }
    struct W2 { float x, y; };
    // a [16][ncols][4] matrix (k0, k2, k1, k3 per quad) and the first columns of the two 32-column tiles this wave multiplies now
    struct Mat { const float* w; int ncols, c0, c1; };
    // (buffer loads: the matrix as a resource, the quad / tile as a SCALAR offset, the lane's constant 16 j + 8 h -- no per-lane 64-bit addresses)
    const int loff = j * 16 + h * 8;
    auto wq = [&](const Mat& M, int q, int t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)M.w, 0, 0x7fffffff, 0x00020000);
        return __builtin_bit_cast(W2, __builtin_amdgcn_raw_buffer_load_b64(rs, loff, (q * M.ncols + (t ? M.c1 : M.c0)) * 16, 0));
    };
    // The weights come through a ring of NPF quads that runs ACROSS the GEMMs: the last NPF steps of one request the first quads of the next
    // (`nxt`), so no GEMM starts with an exposed L2 round trip -- the waves of a launch run the same phases at the same time, nobody else
    // on the SIMD would cover it.  `hook` (the next phase's bias / |C|^2 / codeword requests) goes out before the look-ahead loads: loads
    // return in order, and the wait for them leaves the ring in flight.
    constexpr int NPF = 4;
    W2 ring[NPF][2];
    auto gemm2 = [&](const Mat& cur, const Mat& nxt, const float (&m)[32], v16f (&acc)[2], auto&& hook) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const W2 A0 = ring[q % NPF][0], A1 = ring[q % NPF][1];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, m[2 * q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, m[2 * q], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, m[2 * q + 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, m[2 * q + 1], acc[1], 0, 0, 0);
            if (q == 16 - NPF - 1) hook();
            if (q + NPF < 16) { ring[q % NPF][0] = wq(cur, q + NPF, 0); ring[q % NPF][1] = wq(cur, q + NPF, 1); }
            else { ring[q % NPF][0] = wq(nxt, q + NPF - 16, 0); ring[q % NPF][1] = wq(nxt, q + NPF - 16, 1); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // register r of tile t = column (t ? c1 : c0) + 8 (r >> 2) + 4 h + (r & 3): the four float4 of a per-column vector this lane's registers need
    auto vec16 = [&](const float* p, int c0, int c1, v16f (&o)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f v = *(const v4f*)(p + (t ? c1 : c0) + 8 * q + 4 * h);
                o[t][4 * q] = v[0]; o[t][4 * q + 1] = v[1]; o[t][4 * q + 2] = v[2]; o[t][4 * q + 3] = v[3];
            }
    };
    // C layout -> the wave's LDS rows, (k0, k2, k4, k6, k1, k3, k5, k7) inside every group of eight channels; and back as B operands
    auto to_lds = [&](const v16f (&acc)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* g = row + (4 * t + q) * 8 + 2 * h;
                *(float2*)g = make_float2(acc[t][4 * q], acc[t][4 * q + 2]);
                *(float2*)(g + 4) = make_float2(acc[t][4 * q + 1], acc[t][4 * q + 3]);
            }
    };
    auto from_lds = [&](float (&m)[32]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const v4f v = *(const v4f*)(row + g * 8 + 4 * h);
            m[4 * g] = v[0]; m[4 * g + 1] = v[1]; m[4 * g + 2] = v[2]; m[4 * g + 3] = v[3];
        }
    };

    v16f acc[2], nv[2];                                        // nv: the next phase's bias / |C|^2, requested one phase ahead
    const int npair = (a.ke + 63) >> 6, ppseg = a.segs == 1 ? npair : (a.kc >> 6);
    auto dist_mat = [&](const float* cbp, int P) __attribute__((always_inline)) {
        return Mat{cbp, a.ke, 64 * P, 64 * P + 32 < a.ke ? 64 * P + 32 : 64 * P};      // (an odd last tile: tile 0 once more, its results unused)
    };
    {
        const Mat first{a.lvl[0], D, 0, 32};
#pragma unroll
        for (int q = 0; q < NPF; ++q) { ring[q][0] = wq(first, q, 0); ring[q][1] = wq(first, q, 1); }
        vec16(a.lvl[0] + D * D, 0, 32, nv);
    }
    for (int l = 0; l < a.levels; ++l) {
        const float* W = a.lvl[l];
        const float* stage_w = W;
        const float* stage_b = stage_w + D * D;
        const float* qhead_w = stage_b + D;
        const float* qhead_b = qhead_w + D * D;
        const float* lhead_w = qhead_b + D;
        const float* lhead_b = lhead_w + D * D;
        const float* cbp = lhead_b + D;                       // [16][ke][4]
        const float* cb = cbp + (size_t)D * a.ke;             // [ke][64]
        const float* c2 = cb + (size_t)a.ke * D;              // [ke]
        const bool last = l + 1 == a.levels;
        (void)stage_b;

        // ---- z = stage(x) ----------------------------------------------------------------------------------------------------------------
        acc[0] = nv[0]; acc[1] = nv[1];
        gemm2(Mat{stage_w, D, 0, 32}, Mat{qhead_w, D, 0, 32}, xq, acc, [&]() __attribute__((always_inline)) { vec16(qhead_b, 0, 32, nv); });
        to_lds(acc);
        from_lds(z);
        // ---- q = qhead(z) ----------------------------------------------------------------------------------------------------------------
        acc[0] = nv[0]; acc[1] = nv[1];
        {
            const Mat d0 = dist_mat(cbp, 0);
            gemm2(Mat{qhead_w, D, 0, 32}, d0, z, acc, [&]() __attribute__((always_inline)) { vec16(c2, d0.c0, d0.c1, nv); });
        }
        to_lds(acc);
        from_lds(xq);                                          // xq = q from here to the residual
        // |q_s|^2: ONE ascending fma chain over the segment's 64 / segs dims per (segment, cell); half-wave h runs the segments
        // h * segs / 2 ... (one segment: both run it)
        float x2s[4];
        {
            const int nseg = a.segs, per = nseg == 1 ? 1 : nseg >> 1, g8 = 8 / nseg;      // segments per half-wave; groups of eight dims per segment
            float mine[2] = {0.0f, 0.0f};
#pragma unroll
            for (int sgi = 0; sgi < 2; ++sgi) {
                if (sgi < per) {
                    const int sg = nseg == 1 ? 0 : h * per + sgi;
                    float s2 = 0.0f;
                    for (int g = sg * g8; g < (sg + 1) * g8; ++g) {
                        const v4f e = *(const v4f*)(row + g * 8), o = *(const v4f*)(row + g * 8 + 4);
                        s2 = fmaf(e[0], e[0], s2); s2 = fmaf(o[0], o[0], s2); s2 = fmaf(e[1], e[1], s2); s2 = fmaf(o[1], o[1], s2);
                        s2 = fmaf(e[2], e[2], s2); s2 = fmaf(o[2], o[2], s2); s2 = fmaf(e[3], e[3], s2); s2 = fmaf(o[3], o[3], s2);
                    }
                    mine[sgi] = s2;
                }
            }
            const float o0 = __shfl_xor(mine[0], 32), o1 = __shfl_xor(mine[1], 32);
            if (nseg == 1) { x2s[0] = h ? o0 : mine[0]; x2s[1] = x2s[2] = x2s[3] = 0.0f; }
            else if (nseg == 2) { x2s[0] = h ? o0 : mine[0]; x2s[1] = h ? mine[0] : o0; x2s[2] = x2s[3] = 0.0f; }
            else { x2s[0] = h ? o0 : mine[0]; x2s[1] = h ? o1 : mine[1]; x2s[2] = h ? mine[0] : o0; x2s[3] = h ? mine[1] : o1; }
        }
        // ---- distances to the extended codebook's rows, 64 per step, the running first-argmin inside the lane; a segment's argmin closes
        //      with its last step (kc % 64 == 0 when segs > 1; one segment: an odd last 32-row tile is skipped) ------------------------------
        float best = INFINITY;
        int bc = 0;
        int rows_sel[4] = {0, 0, 0, 0};                        // per segment: the chosen row of the extended codebook
#pragma unroll 1
        for (int P = 0; P < npair; ++P) {
            const int seg = a.segs == 1 ? 0 : P / ppseg;
            const bool two = 64 * P + 32 < a.ke;
            const float x2 = seg == 0 ? x2s[0] : (seg == 1 ? x2s[1] : (seg == 2 ? x2s[2] : x2s[3]));
            v16f c2t[2];
            c2t[0] = nv[0]; c2t[1] = nv[1];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            const bool more = P + 1 < npair;
            const Mat nxt = more ? dist_mat(cbp, P + 1) : (last ? dist_mat(cbp, P) : Mat{lhead_w, D, 0, 32});
            gemm2(dist_mat(cbp, P), nxt, xq, acc, [&]() __attribute__((always_inline)) {
                if (more) vec16(c2, nxt.c0, nxt.c1, nv);
                else if (!last) vec16(lhead_b, 0, 32, nv);
            });
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) {                 // rows ascend with (t, r) inside a lane: a strict < keeps the first
                    const float d = (x2 + c2t[t][r]) - 2.0f * acc[t][r];
                    const int code = 64 * P + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
                    const bool lt = d < best;
                    best = lt ? d : best;
                    bc = lt ? code : bc;
                }
            }
            if (a.segs == 1 ? P + 1 == npair : (P + 1) % ppseg == 0) {       // the segment's last step: the two half-waves' minima, ties to the lower row
                const float od = __shfl_xor(best, 32);
                const int oc = __shfl_xor(bc, 32);
                if (od < best || (od == best && oc < bc)) bc = oc;
                if (h == 0 && m0 + j < a.M) a.codes[((size_t)l * a.segs + seg) * a.M + m0 + j] = (uint8_t)(bc - seg * a.kc);
                if (seg == 0) rows_sel[0] = bc; else if (seg == 1) rows_sel[1] = bc; else if (seg == 2) rows_sel[2] = bc; else rows_sel[3] = bc;
                best = INFINITY; bc = 0;
            }
        }
        if (last) break;
        // ---- x <- lhead(z) - C[row] ------------------------------------------------------------------------------------------------------
        acc[0] = nv[0]; acc[1] = nv[1];
        v16f cv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * t + 8 * q + 4 * h, sg = (ch * a.segs) >> 6;
                const int rsel = sg == 0 ? rows_sel[0] : (sg == 1 ? rows_sel[1] : (sg == 2 ? rows_sel[2] : rows_sel[3]));
                const v4f v = *(const v4f*)(cb + (size_t)rsel * D + ch);
                cv[t][4 * q] = v[0]; cv[t][4 * q + 1] = v[1]; cv[t][4 * q + 2] = v[2]; cv[t][4 * q + 3] = v[3];
            }
        gemm2(Mat{lhead_w, D, 0, 32}, Mat{a.lvl[l + 1], D, 0, 32}, z, acc, [&]() __attribute__((always_inline)) { vec16(a.lvl[l + 1] + D * D, 0, 32, nv); });
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] - cv[t][r];
        to_lds(acc);
        from_lds(xq);
    }
}

__global__ void codebook64_c2_kernel(const float* __restrict__ cb, int kc, float* __restrict__ c2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= kc) return;
    float acc = 0.0f;
    for (int i = 0; i < D; ++i) { const float v = cb[(size_t)k * D + i]; acc = fmaf(v, v, acc); }
    c2[k] = acc;
}

}  // namespace
}  // namespace qv2x

extern "C" int64_t qv2x_codebook64_level_floats(int kc) { return 3LL * (64 * 64 + 64) + 2LL * 64 * kc + kc; }

extern "C" int qv2x_codebook64_c2_f32(const float* codebook, int kc, float* c2, void* stream) {
    using namespace qv2x;
    if (!codebook || !c2 || kc <= 0) return fail(QV2X_EINVAL, "qv2x_codebook64_c2_f32: bad arguments");
    codebook64_c2_kernel<<<(kc + 63) / 64, 64, 0, (hipStream_t)stream>>>(codebook, kc, c2);
    return hip_check(hipGetLastError(), "qv2x_codebook64_c2_f32 launch");
}

static int encode64_launch(const qv2x_encode_desc* d, int cin_total, const int8_t* in, const float* in_f32, const float* const* level_weights,
                           uint8_t* codes, void* stream) {
    using namespace qv2x;
    if (!d || (!in && !in_f32) || !level_weights || !codes) return fail(QV2X_EINVAL, "qv2x_codebook_encode64_f32: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 4) return fail(QV2X_EINVAL, "qv2x_codebook_encode64_f32: bad shape");
    const int segs = d->segs ? d->segs : 1;
    if ((segs != 1 && segs != 2 && segs != 4) || d->kc < 32 || d->kc > 256 || d->kc % 32 || (segs > 1 && d->kc % 64) || segs * d->kc > 32 * MAX_TILES)
        return fail(QV2X_EINVAL, "qv2x_codebook_encode64_f32: seg_num 1 | 2 | 4, dict_size a multiple of 32 up to 256 (of 64 with seg_num > 1; "
                                 "seg_num * dict_size <= %d): got dict_size %d, seg_num %d", 32 * MAX_TILES, d->kc, d->segs);
    if (cin_total < 64 || cin_total % 16 || ((uintptr_t)in & 15) || ((uintptr_t)in_f32 & 15))
        return fail(QV2X_EALIGN, "qv2x_codebook_encode64_f32: >= 64 channels (%% 16), 16-byte aligned map");
    Enc64Args a;
    a.in = in; a.in_f32 = in_f32; a.codes = codes; a.n = d->n; a.h = d->h; a.w = d->w; a.cin_total = cin_total; a.levels = d->levels; a.kc = d->kc;
    a.segs = segs; a.ke = segs * d->kc;
    a.ax = 128 - d->in_zx; a.dx = d->in_delta; a.M = d->n * d->h * d->w;
    for (int l = 0; l < 4; ++l) {
        a.lvl[l] = l < d->levels ? level_weights[l] : nullptr;
        if (l < d->levels && (!a.lvl[l] || ((uintptr_t)a.lvl[l] & 15))) return fail(QV2X_EALIGN, "qv2x_codebook_encode64_f32: level %d weights null or unaligned", l);
    }
    // a wave per 32 cells from 2048 cells on (round 5: 172 -> see DESIGN.md 3 / profiles/r05_deconv_ps.log); below, the workgroup form
    if (a.M >= 2048) {
        const unsigned grid = (unsigned)((a.M + 31) / 32);
        if (in_f32) codebook_encode64_wave_kernel<true><<<grid, 64, 0, (hipStream_t)stream>>>(a);
        else codebook_encode64_wave_kernel<false><<<grid, 64, 0, (hipStream_t)stream>>>(a);
    } else {
        codebook_encode64_kernel<<<(a.M + ER - 1) / ER, 256, 0, (hipStream_t)stream>>>(a);
    }
    return hip_check(hipGetLastError(), "qv2x_codebook_encode64_f32 launch");
}

extern "C" int qv2x_codebook_encode64_f32(const qv2x_encode_desc* d, int cin_total, const int8_t* in, const float* const* level_weights,
                                          uint8_t* codes, void* stream) {
    return encode64_launch(d, cin_total, in, nullptr, level_weights, codes, stream);
}

extern "C" int qv2x_codebook_encode64_f32in(const qv2x_encode_desc* d, int cin_total, const float* in, const float* const* level_weights,
                                            uint8_t* codes, void* stream) {
    return encode64_launch(d, cin_total, nullptr, in, level_weights, codes, stream);
}
