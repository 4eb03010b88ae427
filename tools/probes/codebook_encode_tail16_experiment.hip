// NOT SHIPPED -- an experiment kept with its numbers (profiles/r06_tail16_experiment.log): bit-exact (tests/test_hip_encode_two_stage.py ran green on it)
// and slower than the 32-cell remainder form.  To rebuild it: copy into quantv2x_amd/csrc/, declare encode_list_tail16_launch in codebook_encode.h and call it
// from qv2x_codebook_encode_listed_f32 in place of codebook_encode_list_tail_kernel.
// a6, stage 2 of the two-stage EXACT encode, the REMAINDER of the list in tiles of SIXTEEN cells (round 6).
//
// The workgroup form of codebook_encode.hip gives 32 listed cells to an 8-wave workgroup: one frame's ~100 tiles sit on ~100 of the 256 CUs
// and the launch lasts as long as ONE chain of eleven GEMMs on one CU -- 97 us, 0.74 of what that CU's matrix pipe needs for 32 rows.  Here
// a workgroup takes 16 cells on v_mfma_f32_16x16x4_f32 (1024 MACs in 8 passes: the same rate as the 32x32x2 form, half the rows): twice
// the workgroups, half the chain.  Same arithmetic per (cell, output): every dot product an ascending-k fp32 fma chain with acc0 = bias
// (the 16x16x4 instruction adds its four k in order -- profiles/r01_mfma_probe.log), |q|^2 as four 64-long chains combined (p0 + p1) +
// (p2 + p3), d = (|q|^2 + |C|^2) - 2 q.C, first argmin, x <- lhead(z) - C[code]: bit for bit the indices of encode_rows<32> (and of the
// wave form and the oracle): tests/test_hip_encode_two_stage.py runs every case with either remainder form.
//
// Operand maps of v_mfma_f32_16x16x4_f32: A lane l = (row l & 15, k l >> 4), B lane l = (k l >> 4, column l & 15), D register r of lane l =
// (row 4 (l >> 4) + r, column l & 15).  The weights stay in the blobs of the other forms -- [K/4][col][k0, k2, k1, k3] -- and a lane picks
// its k out of the quad (one dword per MFMA, 16 columns x 16 B contiguous per instruction); the activation tiles in LDS keep, inside
// every 16 k, the order k = g, 4 + g, 8 + g, 12 + g contiguous for g = 0..3, so the lane of k-group g reads ONE float4 per four MFMAs.
// seg_num 1, one round of code tiles (ke <= 128): the models the two-stage encode takes.
#include "codebook_encode.h"

namespace qv2x {
namespace {

constexpr int D = 256;
constexpr int R16 = 16;
#ifndef QV2X_T16_PF
#define QV2X_T16_PF 8
#endif
constexpr int PF = QV2X_T16_PF;     // weight quads per register set (two sets: one in flight while the other feeds the MFMAs)
constexpr int LDF16 = 260;          // LDS row stride in floats: the 16 rows of a k-group start in 16 distinct 16-byte slots
__device__ __forceinline__ int pos16(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }
__device__ __forceinline__ int quad_slot(int g) { return ((g & 1) << 1) | (g >> 1); }       // where k = 4 q + g sits in a packed quad (k0, k2, k1, k3)

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// out[16 rows][cols 32 wave + 16 t + (lane & 15)] = in[16][256] . W^T, acc0 = bias; W packed [64][256][4] floats.
struct Head16 { float s0[PF][2]; float b[2]; };
__device__ __forceinline__ Head16 gemm_head16(const float* __restrict__ wp, const float* __restrict__ bias, int wave, int lane) {
    Head16 h;
    const float* wl = wp + (size_t)(wave * 32 + (lane & 15)) * 4 + quad_slot(lane >> 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) h.b[t] = bias[wave * 32 + 16 * t + (lane & 15)];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) h.s0[j][t] = wl[(size_t)j * D * 4 + t * 64];
    return h;
}

__device__ __forceinline__ void gemm16(const float* __restrict__ src, const float* __restrict__ wp, const Head16& head, int wave, int lane, v4f (&acc)[2]) {
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = head.b[t];
    const float* wl = wp + (size_t)(wave * 32 + (lane & 15)) * 4 + quad_slot(g);
    const float* al = src + (lane & 15) * LDF16 + 4 * g;
    auto loadB = [&](float (&dst)[PF][2], int q0) {                    // quads q0 .. q0 + PF - 1 (past the last one: the next section of the blob)
#pragma unroll
        for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) dst[j][t] = wl[(size_t)(q0 + j) * D * 4 + t * 64];
    };
    auto compute = [&](const float (&b)[PF][2], int q0) {
#pragma unroll
        for (int j4 = 0; j4 < PF; j4 += 4) {
            const v4f a4 = *(const v4f*)(al + 4 * (q0 + j4));           // k = 4 (q0 + j4 + j) + g, j = 0..3
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b[j4 + j][t], acc[t], 0, 0, 0);
        }
    };
    float s0[PF][2], s1[PF][2];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) s0[j][t] = head.s0[j][t];
    for (int q0 = 0; q0 < 64; q0 += 2 * PF) {
        loadB(s1, q0 + PF);
        __builtin_amdgcn_sched_barrier(0);
        compute(s0, q0);
        __builtin_amdgcn_sched_barrier(0);
        loadB(s0, q0 + 2 * PF);
        __builtin_amdgcn_sched_barrier(0);
        compute(s1, q0 + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the whole chain for the listed cells lst[i0 .. i0 + 16) (entries past n_listed: the last one again, nothing stored)
__device__ __forceinline__ void encode_rows16(const EncArgs& a, const int i0, float* __restrict__ smem, const unsigned* __restrict__ lst, const int n_listed) {
    float* bufA = smem;                       // x, then q, then the next x
    float* bufB = smem + R16 * LDF16;         // z
    float* x2 = smem + 2 * R16 * LDF16;       // [16]
    float* pval = x2 + R16;                   // [4][16]
    unsigned long long* pkey = (unsigned long long*)(pval + 4 * R16);      // [8 tiles of 16 codes][16]
    int* code_s = (int*)(pkey + 8 * R16);     // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    {   // 16 rows x 256 channels from the i8 map: thread = (row, 8 channels)
        const int row = tid >> 5, part = tid & 31;
        int m = (int)lst[i0 + row < n_listed ? i0 + row : n_listed - 1];
        m = m < a.M ? m : a.M - 1;
        const int img = m / (a.h * a.w), rem = m - img * (a.h * a.w);
        const int y = rem / a.w, x = rem - y * a.w;
        const size_t pixel = (size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1;
        const int2 raw = *(const int2*)(a.in + pixel * D + part * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int word = e < 4 ? raw.x : raw.y;
            const int xs = (word << (24 - (e & 3) * 8)) >> 24;
            bufA[row * LDF16 + pos16(part * 8 + e)] = (float)(xs + a.ax) * a.dx;
        }
    }
    Head16 head = gemm_head16(a.lvl[0], a.lvl[0] + D * D, wave, lane);
    lds_barrier();

    v4f acc[2];
    for (int l = 0; l < a.levels; ++l) {
        const float* stage_w = a.lvl[l];
        const float* stage_b = stage_w + D * D;
        const float* qhead_w = stage_b + D;
        const float* qhead_b = qhead_w + D * D;
        const float* lhead_w = qhead_b + D;
        const float* lhead_b = lhead_w + D * D;
        const float* cbp = lhead_b + D;                       // [64][ke][4]
        const float* cb = cbp + (size_t)D * a.ke;             // [ke][256]
        const float* c2 = cb + (size_t)a.ke * D;              // [ke]
        auto store = [&](float* dst) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(4 * g + r) * LDF16 + pos16(wave * 32 + 16 * t + l15)] = acc[t][r];
        };
        gemm16(bufA, stage_w, head, wave, lane, acc);                              // z = stage(x)
        head = gemm_head16(qhead_w, qhead_b, wave, lane);
        store(bufB);
        lds_barrier();
        gemm16(bufB, qhead_w, head, wave, lane, acc);                              // q = qhead(z)
        if (l + 1 < a.levels) head = gemm_head16(lhead_w, lhead_b, wave, lane);    // used after the argmin
        store(bufA);
        lds_barrier();

        if (tid < 4 * R16) {   // |q|^2: four 64-long ascending fma chains per row; thread = (chain tid / 16, row tid % 16)
            const int row = tid & 15, part = tid >> 4;
            const float* qr = bufA + row * LDF16;
            float s = 0.0f;
            for (int k = part * 64; k < part * 64 + 64; ++k) {
                const float v = qr[pos16(k)];
                s = fmaf(v, v, s);
            }
            pval[part * R16 + row] = s;
        }
        // the distance tile of this wave: 16 codes.  Its first codebook quads and |C|^2 are requested ahead of the two barriers.
        const int ntile = a.ke >> 4;
        const int code = wave * 16 + l15;
        const float* cl = cbp + (size_t)code * 4 + quad_slot(g);
        float dh[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) dh[j] = 0.0f;
        float c2v = 0.0f;
        if (wave < ntile) {
#pragma unroll
            for (int j = 0; j < PF; ++j) dh[j] = cl[(size_t)j * a.ke * 4];
            c2v = c2[code];
        }
        lds_barrier();
        if (tid < R16) x2[tid] = (pval[tid] + pval[R16 + tid]) + (pval[2 * R16 + tid] + pval[3 * R16 + tid]);
        lds_barrier();

        if (wave < ntile) {
            v4f dacc = {0.f, 0.f, 0.f, 0.f};
            const float* al = bufA + l15 * LDF16 + 4 * g;
            auto loadC = [&](float (&dst)[PF], int q0) {
#pragma unroll
                for (int j = 0; j < PF; ++j) dst[j] = cl[(size_t)(q0 + j) * a.ke * 4];      // past the end: the [ke][256] copy
            };
            auto dist = [&](const float (&b)[PF], int q0) {
#pragma unroll
                for (int j4 = 0; j4 < PF; j4 += 4) {
                    const v4f a4 = *(const v4f*)(al + 4 * (q0 + j4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b[j4 + j], dacc, 0, 0, 0);
                }
            };
            float s0[PF], s1[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) s0[j] = dh[j];
            for (int q0 = 0; q0 < 64; q0 += 2 * PF) {
                loadC(s1, q0 + PF);
                __builtin_amdgcn_sched_barrier(0);
                dist(s0, q0);
                __builtin_amdgcn_sched_barrier(0);
                loadC(s0, q0 + 2 * PF);
                __builtin_amdgcn_sched_barrier(0);
                dist(s1, q0 + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (distance, code) minimum of each of this lane's four rows over the 16 codes of its DPP row: the butterfly of codebook_encode.hip
            // without the step across rows; the FIRST lane holding the minimum = the lowest code (the reference's first-argmin).
            unsigned resk = 0xffffffffu;
            int resc = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = (x2[4 * g + r] + c2v) - 2.0f * dacc[r];
                unsigned b = __builtin_bit_cast(unsigned, d);
                b ^= (unsigned)((int)b >> 31) | 0x80000000u;
                unsigned m = b, o;
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xf, 0xf, false); m = o < m ? o : m;      // quad_perm [1,0,3,2]
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xf, 0xf, false); m = o < m ? o : m;      // quad_perm [2,3,0,1]
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x141, 0xf, 0xf, false); m = o < m ? o : m;     // row_half_mirror
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x140, 0xf, 0xf, false); m = o < m ? o : m;     // row_mirror
                const unsigned long long eq = __builtin_amdgcn_ballot_w64(b == m);
                const int c = __builtin_ctz((unsigned)(eq >> (16 * g)) & 0xffffu);
                const bool mine = l15 == r;
                resk = mine ? m : resk;
                resc = mine ? c : resc;
            }
            if (l15 < 4) pkey[wave * R16 + 4 * g + l15] = ((unsigned long long)resk << 32) | (unsigned)(wave * 16 + resc);
        }
        lds_barrier();
        if (tid < R16) {
            unsigned long long bk = pkey[tid];
            for (int w = 1; w < ntile; ++w) {
                const unsigned long long ok = pkey[w * R16 + tid];
                bk = ok < bk ? ok : bk;                                // code tiles ascend: ties still go to the lower code
            }
            const int bi = (int)(unsigned)bk;
            code_s[tid] = bi;
            if (i0 + tid < n_listed) a.codes[(size_t)l * a.M + lst[i0 + tid]] = (uint8_t)bi;
        }
        lds_barrier();

        if (l + 1 < a.levels) {      // x <- lhead(z) - C[code]
            float cv[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[t][r] = cb[(size_t)code_s[4 * g + r] * D + wave * 32 + 16 * t + l15];
            gemm16(bufB, lhead_w, head, wave, lane, acc);
            head = gemm_head16(a.lvl[l + 1], a.lvl[l + 1] + D * D, wave, lane);     // the next level's stage
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) bufA[(4 * g + r) * LDF16 + pos16(wave * 32 + 16 * t + l15)] = acc[t][r] - cv[t][r];
            lds_barrier();
        }
    }
}

constexpr int SMEM16 = 2 * R16 * LDF16 + R16 + 4 * R16 + 2 * 8 * R16 + R16 + 16;

// the tiles [plan.full, plan.total) of 32 listed cells (codebook_encode.h:list_plan), each as two halves of 16.
// PAD: extra LDS floats -- the launches of one or two frames (a few hundred workgroups at most) take the form that fills a CU with ONE workgroup,
// so that the dispatcher spreads them over the chip instead of pairing them on half of it.
template <int PAD>
__global__ __launch_bounds__(512, 2) void codebook_encode_list_tail16_kernel(const EncArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM16 + PAD];
    const ListPlan plan = list_plan(a);
    const int halves = 2 * (plan.total - plan.full);
    for (int t = (int)blockIdx.x; t < halves; t += (int)gridDim.x) {
        int cls, i0;
        list_tile(plan, plan.full + (t >> 1), cls, i0);
        i0 += 16 * (t & 1);
        if (i0 < plan.n[cls]) encode_rows16(a, i0, smem, a.list + (size_t)cls * a.M, plan.n[cls]);      // (uniform; a tile's second half may be empty)
        __syncthreads();
    }
}

}  // namespace

int encode_list_tail16_launch(const EncArgs& a, int grid, bool spread, hipStream_t st) {
    if (spread) codebook_encode_list_tail16_kernel<12288><<<grid, 512, 0, st>>>(a);
    else codebook_encode_list_tail16_kernel<0><<<grid, 512, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_listed_f32 launch (16-cell remainder)");
}

}  // namespace qv2x
