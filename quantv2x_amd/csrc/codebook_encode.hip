// a6: UMGMQuantizer.encode (codebook.py:330-337 -> :231-239 -> :106-131), m = 1, D = 256, fused over all
// residual levels on v_mfma_f32_32x32x2_f32.
//
// Per level:  z = stage(x);  q = qhead(z);  d_k = (|q|^2 + |C_k|^2) - 2 (q . C_k);  code = first argmin_k;
//             x <- lhead(z) - C[code]          (reference op order; every dot product is an ascending-k fp32
//             fma chain with acc0 = bias, which is what the f32 MFMA evaluates -- bit-exact against
//             oracle/qv2x_oracle.c:orc_codebook_encode).
// A workgroup (8 waves) owns 64 BEV cells: x / z / q live in two LDS buffers (row stride 260 floats: ds_read_b128 of
// a 16-lane group lands on 16 distinct 16-byte slots), the 256 x 256 weight matrices stream from L2 as
// [K/4][col][k0,k2,k1,k3] (coalesced float2 per half-wave), prefetched four k-quads ahead.  Wave w owns output columns
// [32w, 32w + 32) for all 64 rows; for the distance GEMM one 32 x 32 (rows x codes) tile.
// The argmin runs on the accumulator registers: half-wave butterfly on (distance, index) with ties to the lower
// index, then a 4-way combine across waves through LDS.
#include <cstdlib>

#include <atomic>

#include "codebook_encode.h"

namespace qv2x {

#ifdef QV2X_ENC_TRACE   // dev build (tools/enc_block_trace.py): when and on which CU every workgroup ran
__device__ unsigned long long g_enc_blk[4096][4];
#endif
#ifdef QV2X_ENC_FINE    // dev build (tools/enc_fine.py): s_memtime stamps of thread 0 at the phase boundaries of every level (12 per level)
__device__ long long g_enc_fine[2048][40];
#define EFINE(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_enc_fine[blockIdx.x][12 * l + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EFINE(k) do { } while (0)
#endif

// ER = rows per workgroup (template parameter of the kernel), ERT = ER / 32 MFMA row tiles per wave:
//   32  two workgroups share a CU (4 waves per SIMD): the finer grain -- one frame is 1100 workgroups on 512 slots;
//   64  one workgroup per CU with two accumulator chains per wave: every weight element fetched feeds two MFMAs and a wave waits for
//       LDS / L2 once per 16 MFMAs instead of 8 -- 11.2 against 11.8 ms per batch of 32 frames (0.795 against 0.76 of the f32 peak), but
//       550 workgroups on 256 slots for one frame (519 against 475 us): launches of two frames and more take it.
constexpr int LDF = 258;            // LDS row stride in floats: 32 rows x ds_read_b64 land on 32 distinct bank pairs
// Within every group of four consecutive k the LDS tiles and the packed weights are stored in the order (k0, k2, k1, k3):
// the half-wave that feeds MFMA k-parity p reads ONE float2 = (k_p, k_{p+2}) -- no per-MFMA select instructions, so the
// dependent MFMAs issue back to back (an extra VALU between two MFMAs on one accumulator costs ~45 cycles on gfx950).
__device__ __forceinline__ int kpos(int c) { return (c & ~3) | ((c & 1) << 1) | ((c >> 1) & 1); }
constexpr int D = 256;
#ifndef QV2X_ENC_LDS_PAD        // dev builds: extra LDS floats per workgroup (occupancy experiments)
#define QV2X_ENC_LDS_PAD 0
#endif

// Workgroup barrier for LDS hand-offs only.  __syncthreads() is fence + barrier, and the fence drains vmcnt: the next phase's weight
// k-quads, requested ahead of the barrier on purpose, would have to land before the barrier instead of behind it.  Nothing in this
// kernel passes data between waves through global memory, so ordering the LDS traffic is all a barrier has to do here.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}


// out[64 rows][cols 32*wave .. +32) = in[64][256] . W^T, acc0 = bias.  W packed [64][256][4].  Eight waves split
// the 256 output columns, so every weight element is fetched by exactly one wave of the workgroup; two waves share
// a SIMD and cover each other's LDS / L2 latency (one f32 MFMA occupies the pipe for 64 cycles).
// First four k-quads of a weight matrix + the bias of this lane's output column, requested ahead of the barrier that
// precedes the GEMM so that its L2 latency overlaps the previous phase.
// out[rows][cols 32*wave .. +32) = in[rows][256] . W^T, acc0 = bias.  Eight waves split the 256 output columns, so
// every weight element is fetched by exactly one wave of the workgroup.  The weights of the next four k-quads (one
// float2 per lane each) are requested from L2 while the 8 MFMAs of the current four run; two static register sets,
// so the in-order vmcnt wait at a set's first use leaves the other set's loads in flight (sched_barrier pins the
// request block ahead of the MFMA block: hipcc otherwise sinks every load to its first use).  Reads past the last
// k-quad hit the next section of the weight blob.
// the first four k-quads of a weight matrix + the bias of this lane's column: requested BEFORE the barrier that precedes the GEMM, so
// the L2 round trip overlaps the previous phase's tile store and the barrier wait
struct GemmHead { float2 s0[4]; float b; };
__device__ __forceinline__ GemmHead gemm_head(const float2* __restrict__ wp, const float* __restrict__ bias, int wave, int lane) {
    const int par = lane >> 5, col = wave * 32 + (lane & 31);
    GemmHead h;
    h.b = bias[col];
    const float2* wl = wp + (size_t)col * 2 + par;
#pragma unroll
    for (int t = 0; t < 4; ++t) h.s0[t] = wl[(size_t)t * D * 2];
    return h;
}

template <int ERT>
__device__ __forceinline__ void gemm_rows_x32(const float* __restrict__ src, const float2* __restrict__ wp, const GemmHead& head,
                                              int wave, int lane, v16f (&acc)[ERT]) {
    const int par = lane >> 5, col = wave * 32 + (lane & 31);
    const float b = head.b;
#pragma unroll
    for (int i = 0; i < ERT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = b;
    const float2* wl = wp + (size_t)col * 2 + par;
    const float* al = src + (lane & 31) * LDF + 2 * par;
    auto loadB = [&](float2 (&dst)[4], int q0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = wl[(size_t)(q0 + t) * D * 2];
    };
    auto compute4 = [&](const float2 (&bset)[4], int q0) {
        float2 av[4][ERT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < ERT; ++i) av[t][i] = *(const float2*)(al + i * 32 * LDF + (q0 + t) * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < ERT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][i].x, bset[t].x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < ERT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][i].y, bset[t].y, acc[i], 0, 0, 0);
        }
    };
    float2 s0[4], s1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s0[t] = head.s0[t];
    for (int q0 = 0; q0 < 64; q0 += 8) {
        loadB(s1, q0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        compute4(s0, q0);
        __builtin_amdgcn_sched_barrier(0);
        loadB(s0, q0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        compute4(s1, q0 + 4);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// (distance, code) as one sortable 64-bit key: the fp32 bits mapped monotonically onto u32 in the high word, the code in
// the low word.  min(key) = smallest distance, ties to the lower code -- the reference's first-argmin -- with one 64-bit
// compare per step and no short-circuit branches (the (value, index) pair with `||` / `&&` compiled to ~100 instructions
// and four exec-mask branches per accumulator register: 8-25 us per level).  Distances are (x2 + c2) - 2 dot: never -0.
__device__ __forceinline__ unsigned long long dist_key(float d, int code) {
    unsigned b = __builtin_bit_cast(unsigned, d);
    b ^= (unsigned)((int)b >> 31) | 0x80000000u;
    return ((unsigned long long)b << 32) | (unsigned)code;
}

// one step of the key-min reduction: combine with the lane `shift` positions up inside the 16-lane DPP row
template <int ERT>
__device__ __forceinline__ void store_tile(float* __restrict__ dst, int wave, int lane, const v16f (&acc)[ERT]) {
#pragma unroll
    for (int i = 0; i < ERT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dst[(i * 32 + mfma32_row(r, lane)) * LDF + kpos(wave * 32 + (lane & 31))] = acc[i][r];
}

constexpr int ENC_MAX_TILES = 16;   // 32-code tiles of the extended codebook: segs * kc <= 512
template <int ER> constexpr int enc_smem_floats() { return 2 * ER * LDF + 4 * ER + 4 * ER + 2 * ENC_MAX_TILES * ER + 4 * ER + QV2X_ENC_LDS_PAD; }

// the whole encode of the ER rows [m0, m0 + ER) by one 8-wave workgroup; smem: enc_smem_floats<ER>() floats
// EXT = false: one segment and one round of code tiles (ke <= 32 * 8 / ERT: every (m, kc) = (1, <= 128) model) -- the loop over tile rounds
// and the per-segment bookkeeping cost the 32-row form 15 registers and 8 bytes of scratch under its 128-register bound, so they are
// compiled out of the form every V2X-Real / OPV2V attfuse model takes; EXT = true: seg_num > 1 or an extended codebook of several rounds.
// LIST (round 6): the rows are the listed cells a.list[m0 .. m0 + ER) (stage 2 of the two-stage encode), not the cells m0 .. m0 + ER.
template <int ER, bool EXT, bool LIST = false>
__device__ __forceinline__ void encode_rows(const EncArgs& a, const int m0, float* __restrict__ smem, const unsigned* __restrict__ lst = nullptr,
                                            const int n_listed = 0, const int start = 0) {
    constexpr int ERT = ER / 32;
    const int segs = EXT ? a.segs : 1;
    float* bufA = smem;                       // x, then q, then next x
    float* bufB = smem + ER * LDF;            // z
    float* x2 = smem + 2 * ER * LDF;          // [segs <= 4][ER]: |q_s|^2 per segment
    float* pval = x2 + 4 * ER;                // [4][ER]
    unsigned long long* pkey = (unsigned long long*)(pval + 4 * ER);   // [ENC_MAX_TILES][ER] (distance, code) keys of the code tiles
    int* code_s = (int*)(pkey + ENC_MAX_TILES * ER);                   // [segs][ER]: the chosen row of the extended codebook per segment

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef QV2X_ENC_TRACE
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#endif
    {   // ---- load ER rows x 256 channels from the i8 BEV, dequantize ---------------------------------
        constexpr int TPR = 512 / ER;                 // threads per row
        constexpr int CPT = D / TPR;                  // channels per thread (16 for ER = 32)
        const int row = tid / TPR, part = tid % TPR;
        int m = m0 + row;
        if (LIST) m = (int)lst[m < n_listed ? m : n_listed - 1];
        m = m < a.M ? m : a.M - 1;
        const int img = m / (a.h * a.w), rem = m - img * (a.h * a.w);
        const int y = rem / a.w, x = rem - y * a.w;
        const size_t pixel = (size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1;
        const int8_t* src = a.in + pixel * D + part * CPT;
        if (a.in_f32) {
            const float* sf = a.in_f32 + pixel * D + part * CPT;
#pragma unroll
            for (int c = 0; c < CPT / 4; ++c) {
                const v4f v = *(const v4f*)(sf + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) bufA[row * LDF + kpos(part * CPT + c * 4 + e)] = v[e];
            }
        } else
#pragma unroll
        for (int c = 0; c < CPT / 16; ++c) {
            const v4i raw = *(const v4i*)(src + c * 16);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xs = (raw[e >> 2] << (24 - (e & 3) * 8)) >> 24;
                bufA[row * LDF + kpos(part * CPT + c * 16 + e)] = (float)(xs + a.ax) * a.dx;
            }
        }
    }
    GemmHead head = gemm_head((const float2*)a.lvl[0], a.lvl[0] + D * D, wave, lane);      // stage_w / stage_b of level 0
    lds_barrier();

    v16f acc[ERT];
    for (int l = 0; l < a.levels; ++l) {
        const float* W = a.lvl[l];
        const float* stage_w = W;
        const float* stage_b = stage_w + D * D;
        const float* qhead_w = stage_b + D;
        const float* qhead_b = qhead_w + D * D;
        const float* lhead_w = qhead_b + D;
        const float* lhead_b = lhead_w + D * D;
        // (the EXTENDED codebook: ke = segs * kc rows, segment s in dims [s d, (s + 1) d) of its rows, zeros elsewhere)
        const float* cbp = lhead_b + D;                       // [64][ke][4]
        const float* cb = cbp + (size_t)D * a.ke;             // [ke][256]
        const float* c2 = cb + (size_t)a.ke * D;              // [ke]

        EFINE(0);
        gemm_rows_x32(bufA, (const float2*)stage_w, head, wave, lane, acc);         // z = stage(x)
        EFINE(1);
        // LIST: the cells' codes below level `start` are PROVEN (the candidate stage accepted them): their quantization head, |q|^2, distances
        // and argmin are skipped, the stored code is used (as the wave form does: codebook_encode_wave.hip)
        const bool proven = LIST && l < start;                                      // (uniform)
        if (proven) {
            head = gemm_head((const float2*)lhead_w, lhead_b, wave, lane);
            store_tile(bufB, wave, lane, acc);
            if (tid < ER) code_s[tid] = (int)a.codes[(size_t)l * a.M + lst[m0 + tid < n_listed ? m0 + tid : n_listed - 1]];
            lds_barrier();
        } else {
        head = gemm_head((const float2*)qhead_w, qhead_b, wave, lane);
        store_tile(bufB, wave, lane, acc);
        lds_barrier();
        EFINE(2);
        gemm_rows_x32(bufB, (const float2*)qhead_w, head, wave, lane, acc);         // q = qhead(z)
        EFINE(3);
        if (l + 1 < a.levels) head = gemm_head((const float2*)lhead_w, lhead_b, wave, lane);      // used after the argmin
        store_tile(bufA, wave, lane, acc);                                        // x is dead since the barrier above
        lds_barrier();
        EFINE(4);

        if (tid < ER * 4) {   // |q|^2: four 64-wide ascending fma chains per row; thread = (chain = tid / ER, row = tid % ER)
            const int row = tid % ER, part = tid / ER;
            const float2* qr = (const float2*)(bufA + row * LDF + part * 64);
            float s = 0.0f;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {             // one k-quad per step, stored (k0, k2, k1, k3)
                const float2 e = qr[2 * j], o = qr[2 * j + 1];
                s = fmaf(e.x, e.x, s); s = fmaf(o.x, o.x, s); s = fmaf(e.y, e.y, s); s = fmaf(o.y, o.y, s);
            }
            pval[part * ER + row] = s;
        }
        // the distance tile's first codebook k-quads and |C_k|^2, requested ahead of the two barriers of the |q|^2 reduction.
        // Code tiles per round of the workgroup: 8 / ERT (ER = 32: wave -> tile `wave` of row tile 0; ER = 64: wave -> (row tile wave >> 2,
        // tile wave & 3)); an extended codebook of more than one round of tiles (ke > 128 | 256) takes further rounds.
        constexpr int TPR = 8 / ERT;
        const int ct0 = wave % TPR, rt = wave / TPR;
        const int ntile = (a.ke + 31) >> 5;
        float2 dhead[4];
        float c2v = 0.0f;
        if (ct0 < ntile) {
            const int code = ct0 * 32 + (lane & 31);
            const float2* cl = (const float2*)cbp + (size_t)code * 2 + (lane >> 5);
#pragma unroll
            for (int t = 0; t < 4; ++t) dhead[t] = cl[(size_t)t * a.ke * 2];
            c2v = c2[code];
        }
        lds_barrier();
        EFINE(5);
        if (tid < ER) {
            // |q_s|^2 per segment (oracle/qv2x_oracle.c:sumsq_seg): one segment (p0 + p1) + (p2 + p3); two: p0 + p1 | p2 + p3; four: the chains
            const float p0 = pval[tid], p1 = pval[ER + tid], p2 = pval[2 * ER + tid], p3 = pval[3 * ER + tid];
            if (segs == 1) x2[tid] = (p0 + p1) + (p2 + p3);
            else if (segs == 2) { x2[tid] = p0 + p1; x2[ER + tid] = p2 + p3; }
            else { x2[tid] = p0; x2[ER + tid] = p1; x2[2 * ER + tid] = p2; x2[3 * ER + tid] = p3; }
        }
        lds_barrier();
        EFINE(6);

        // ---- distances: wave -> (row tile, 32 codes of the extended codebook), then the argmin per 32-code tile --------------------
#pragma unroll 1
        for (int ct = ct0; ct < ntile; ct += EXT ? TPR : ENC_MAX_TILES) {
            const int par = lane >> 5, code = ct * 32 + (lane & 31);
            v16f dacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[r] = 0.0f;
            const float2* cl = (const float2*)cbp + (size_t)code * 2 + par;
            const float* al = bufA + (rt * 32 + (lane & 31)) * LDF + 2 * par;
            auto loadC = [&](float2 (&dst)[4], int q0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[t] = cl[(size_t)(q0 + t) * a.ke * 2];   // past the end: the [ke][256] copy
            };
            auto dist4 = [&](const float2 (&bset)[4], int q0) {
                float2 av[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) av[t] = *(const float2*)(al + (q0 + t) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, bset[t].x, dacc, 0, 0, 0);
                    dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, bset[t].y, dacc, 0, 0, 0);
                }
            };
            float2 s0[4], s1[4];
            if (ct == ct0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) s0[t] = dhead[t];
            } else {                                                        // a further round: its head and |C|^2 only now
                loadC(s0, 0);
                c2v = c2[code];
            }
            for (int q0 = 0; q0 < 64; q0 += 8) {
                loadC(s1, q0 + 4);
                __builtin_amdgcn_sched_barrier(0);
                dist4(s0, q0);
                __builtin_amdgcn_sched_barrier(0);
                loadC(s0, q0 + 8);
                __builtin_amdgcn_sched_barrier(0);
                dist4(s1, q0 + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
            EFINE(7);
            // (distance, code) minimum of every row over this wave's 32 codes.  The distance's fp32 bits map monotonically onto u32 (as in
            // dist_key); its minimum over the half-wave that holds the row takes four v_min_u32 with DPP operands (quad_perm, quad_perm,
            // row_half_mirror, row_mirror: a butterfly, every lane of a 16-lane row ends with the row's minimum) and one v_permlane16_swap
            // across the two rows; the FIRST lane holding it (= the lowest code: ties go to the lower code, the reference's first-argmin) is
            // a find-first-bit of the equality mask.  ~20 VALU instructions per register against 64 for the 64-bit key chain this
            // replaces, whose 1000 instructions stood between the distance GEMM and the barrier the other four waves wait at
            // (tools/enc_fine.py: 21-28k of a workgroup's 387k ticks per level).  Lane j < 16 of each half-wave collects register j's result.
            unsigned resk = 0xffffffffu;
            int resc = 0;
            const int l31 = lane & 31, hi = lane >> 5;
            const float* x2r = x2 + (segs == 1 ? 0 : (ct * 32) / a.kc) * ER;   // (segs > 1: kc % 64 == 0, a tile lies inside one segment)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + mfma32_row(r, lane);
                const float d = (x2r[row] + c2v) - 2.0f * dacc[r];
                unsigned b = __builtin_bit_cast(unsigned, d);
                b ^= (unsigned)((int)b >> 31) | 0x80000000u;
                unsigned m = b, o;
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xf, 0xf, false); m = o < m ? o : m;      // quad_perm [1,0,3,2]
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xf, 0xf, false); m = o < m ? o : m;      // quad_perm [2,3,0,1]
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x141, 0xf, 0xf, false); m = o < m ? o : m;     // row_half_mirror
                o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x140, 0xf, 0xf, false); m = o < m ? o : m;     // row_mirror
                const auto sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);   // rows 1 | 3 of the first <-> rows 0 | 2 of the second
                m = sw[0] < sw[1] ? sw[0] : sw[1];
                const unsigned long long eq = __builtin_amdgcn_ballot_w64(b == m);
                const int c_lo = __builtin_ctz((unsigned)eq), c_hi = __builtin_ctz((unsigned)(eq >> 32));
                const int c = hi ? c_hi : c_lo;
                const bool mine = l31 == r;
                resk = mine ? m : resk;
                resc = mine ? c : resc;
            }
            if (l31 < 16) pkey[ct * ER + rt * 32 + mfma32_row(l31, lane)] = ((unsigned long long)resk << 32) | (unsigned)(ct * 32 + resc);
        }
        lds_barrier();
        EFINE(8);
        if (tid < ER) {
            const int tps = segs == 1 ? ntile : a.kc >> 5;                  // 32-code tiles per segment
            for (int sg = 0; sg < segs; ++sg) {
                unsigned long long bk = pkey[(sg * tps) * ER + tid];
                for (int wv = 1; wv < tps; ++wv) {
                    const unsigned long long ok = pkey[(sg * tps + wv) * ER + tid];
                    bk = ok < bk ? ok : bk;                            // code tiles ascend: ties still go to the lower code
                }
                const int bi = (int)(unsigned)bk;                      // row of the extended codebook: sg * kc + code
                code_s[sg * ER + tid] = bi;
                if (LIST) {
                    if (m0 + tid < n_listed) a.codes[((size_t)l * segs + sg) * a.M + lst[m0 + tid]] = (uint8_t)(bi - sg * a.kc);
                } else if (m0 + tid < a.m_hi) a.codes[((size_t)l * segs + sg) * a.M + m0 + tid] = (uint8_t)(bi - sg * a.kc);
            }
        }
        lds_barrier();
        EFINE(9);
        }

        if (l + 1 < a.levels) {      // x <- lhead(z) - C[code]
            // the chosen codewords' entries of this lane's column: 16 gathers requested ahead of the GEMM that produces the minuend
            const int col = wave * 32 + (lane & 31);
            const int* cs = code_s + ((wave * segs) >> 3) * ER;      // this wave's 32 columns lie in segment wave * segs / 8
            float cv[ERT][16];
#pragma unroll
            for (int i = 0; i < ERT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) cv[i][r] = cb[(size_t)cs[i * 32 + mfma32_row(r, lane)] * D + col];
            gemm_rows_x32(bufB, (const float2*)lhead_w, head, wave, lane, acc);
            EFINE(10);
            head = gemm_head((const float2*)a.lvl[l + 1], a.lvl[l + 1] + D * D, wave, lane);     // the next level's stage
#pragma unroll
            for (int i = 0; i < ERT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) bufA[(i * 32 + mfma32_row(r, lane)) * LDF + kpos(col)] = acc[i][r] - cv[i][r];
            lds_barrier();
            EFINE(11);
        }
    }
#ifdef QV2X_ENC_TRACE
    if (tid == 0 && blockIdx.x < 4096) {
        g_enc_blk[blockIdx.x][0] = t_start; g_enc_blk[blockIdx.x][1] = __builtin_readcyclecounter();
        g_enc_blk[blockIdx.x][2] = hwid; g_enc_blk[blockIdx.x][3] = xcc;
    }
#endif
}

// static LDS (67.8 / 135.7 KB): no per-device hipFuncSetAttribute state to keep (include/qv2x.h:12)
template <int ER, bool EXT = false>
__global__ __launch_bounds__(512, ER == 32 ? 4 : 2) void codebook_encode_kernel(const EncArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[enc_smem_floats<ER>()];
    encode_rows<ER, EXT>(a, a.m_lo + blockIdx.x * ER, smem);
}

// Stage 2 of the two-stage encode, the REMAINDER of the list: the wave form (codebook_encode_wave.hip, LIST) takes whole rounds of its
// `a.list_slots` persistent waves; what is left -- up to `a.list_tail_max` tiles of 32 cells, e.g. ALL of one frame's ~100 tiles -- runs here
// as 8-wave workgroups, a third of a wave's latency (the same split qv2x_codebook_encode_f32 makes on the host; here both kernels derive it
// from the DEVICE-side counts: codebook_encode.h:list_plan).
__global__ __launch_bounds__(512, 2) void codebook_encode_list_tail_kernel(const EncArgs a) {      // (71 KB of LDS: two per CU whatever the registers)
    __shared__ __attribute__((aligned(16))) float smem[enc_smem_floats<32>()];
    const ListPlan plan = list_plan(a);
    for (int t = plan.full + (int)blockIdx.x; t < plan.total; t += (int)gridDim.x) {
        int cls, i0;
        list_tile(plan, t, cls, i0);                                    // list c: the cells first undecided at level c
        encode_rows<32, false, true>(a, i0, smem, a.list + (size_t)cls * a.M, plan.n[cls], cls);
        __syncthreads();
    }
}

// Launches of one or two frames: whole rounds of 64-row workgroups (one per CU: n64 of them, a multiple of the CU count), then the rows
// that would make a mostly empty round as 32-row workgroups -- which finish in 0.56 of a 64-row workgroup's time alone on their CU.
// One V2X-Real frame: 512 x 64 + 76 x 32 rows, two full rounds and a short one (416 us) instead of 1100 x 32 on 512 slots (478 us).
template <bool EXT>
__global__ __launch_bounds__(512, 2) void codebook_encode_mixed_kernel(const EncArgs a, const int n64) {
    __shared__ __attribute__((aligned(16))) float smem[enc_smem_floats<64>()];
    if ((int)blockIdx.x < n64) encode_rows<64, EXT>(a, a.m_lo + blockIdx.x * 64, smem);
    else encode_rows<32, EXT>(a, a.m_lo + n64 * 64 + ((int)blockIdx.x - n64) * 32, smem);
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

static int encode_launch(const qv2x_encode_desc* d, const int8_t* in, const float* in_f32, const float* const* level_weights, uint8_t* codes, void* stream,
                         int form = 0);

// the wave-per-32-cells form whatever the launch size (qv2x_codebook_encode_f32 takes it from three rounds of the chip on): same codes
extern "C" int qv2x_codebook_encode_wave_f32(const qv2x_encode_desc* d, const int8_t* in, const float* in_f32, const float* const* level_weights,
                                             uint8_t* codes, void* stream) {
    if (!in && !in_f32) return qv2x::fail(QV2X_EINVAL, "qv2x_codebook_encode_wave_f32: null pointer");
    return encode_launch(d, in_f32 ? (const int8_t*)in_f32 : in, in_f32, level_weights, codes, stream, 1);
}

extern "C" int qv2x_codebook_encode_f32(const qv2x_encode_desc* d, const int8_t* in, const float* const* level_weights,
                                        uint8_t* codes, void* stream) {
    return encode_launch(d, in, nullptr, level_weights, codes, stream);
}

extern "C" int qv2x_codebook_encode_f32in(const qv2x_encode_desc* d, const float* in, const float* const* level_weights,
                                          uint8_t* codes, void* stream) {
    if (!in) return qv2x::fail(QV2X_EINVAL, "qv2x_codebook_encode_f32in: null pointer");
    return encode_launch(d, (const int8_t*)in, in, level_weights, codes, stream);
}

// Stage 2 of the two-stage exact encode (encode_two_stage.py, codebook_encode_cand.hip): qv2x_codebook_encode_f32's arithmetic on the cells
// `list[0 .. *list_count)` only -- the count lives on the DEVICE (the candidate stage wrote it), the launch is a fixed number of persistent
// waves, so the pair of launches is capturable in a HIP graph.
extern "C" int qv2x_codebook_encode_listed_f32(const qv2x_encode_desc* d, const int8_t* in, const float* const* level_weights,
                                               const uint32_t* list, const uint32_t* list_count, uint8_t* codes, void* stream) {
    using namespace qv2x;
    if (!d || !in || !level_weights || !codes || !list || !list_count) return fail(QV2X_EINVAL, "qv2x_codebook_encode_listed_f32: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 4) return fail(QV2X_EINVAL, "qv2x_codebook_encode_listed_f32: bad shape");
    if (d->segs > 1) return fail(QV2X_EINVAL, "qv2x_codebook_encode_listed_f32: seg_num 1 only");
    if (d->kc < 32 || d->kc > 256 || d->kc % 32) return fail(QV2X_EINVAL, "qv2x_codebook_encode_listed_f32: dict_size must be a multiple of 32 up to 256 (got %d)", d->kc);
    if ((uintptr_t)in & 15) return fail(QV2X_EALIGN, "qv2x_codebook_encode_listed_f32: in must be 16-byte aligned");
    EncArgs a;
    a.in = in; a.in_f32 = nullptr; a.codes = codes; a.n = d->n; a.h = d->h; a.w = d->w; a.levels = d->levels; a.kc = d->kc;
    a.segs = 1; a.ke = d->kc;
    a.ax = 128 - d->in_zx; a.dx = d->in_delta; a.M = d->n * d->h * d->w; a.m_lo = 0; a.m_hi = a.M;
    a.list = list; a.list_count = list_count;
    for (int l = 0; l < 4; ++l) {
        a.lvl[l] = l < d->levels ? level_weights[l] : nullptr;
        if (l < d->levels && (!a.lvl[l] || ((uintptr_t)a.lvl[l] & 15))) return fail(QV2X_EALIGN, "qv2x_codebook_encode_listed_f32: level %d weights null or unaligned", l);
    }
    if (d->kc > 128) return fail(QV2X_EINVAL, "qv2x_codebook_encode_listed_f32: dict_size <= 128 (the candidate stage's contract)");
    int cus = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    const int tiles = (a.M + 31) / 32;
    int tail_knob = 0;
#ifdef QV2X_DEV_KNOBS                                                  // dev builds only: the largest remainder (tiles) the workgroup form takes
    static const int tail_env = getenv("QV2X_LIST_TAIL") ? atoi(getenv("QV2X_LIST_TAIL")) : 0;
    tail_knob = tail_env;
#endif
    a.list_slots = 4 * cus; a.list_tail_max = tail_knob > 0 ? tail_knob : 2 * cus;
    // whole rounds of the chip's wave slots as persistent waves, the remainder as workgroups: both launches have a fixed size, the split is
    // made on the device from the count (list_plan).  Launches of fewer than two rounds of tiles (one or two frames) are the workgroup form's
    // whatever the count: it finishes ALL 1100 tiles of a frame in 178 us where a round of waves takes ~350, and the wave launch costs 28 us
    // of a one-frame graph even when it finds nothing to do.
    const int grid_tail = 2 * cus;
    if (tiles + 3 >= 2 * a.list_slots) {
        if (int rc = encode_wave_list_launch(a, a.list_slots, (hipStream_t)stream)) return rc;
    } else {
        a.list_slots = 1 << 30; a.list_tail_max = 1 << 30;
    }
    // (the remainder in tiles of SIXTEEN cells on v_mfma_f32_16x16x4_f32 -- twice the workgroups, half the chain -- was built, bit-exact, and
    // slower: 145 against 97 us for one frame, 1164 against 940 us at the batch.  Every workgroup streams the same 2.7 MB of weights, and what
    // the launch waits for is that stream (~3-4 TB/s over the chip whatever the prefetch depth), not the matrix pipe:
    // tools/probes/codebook_encode_tail16_experiment.hip, profiles/r06_tail16_experiment.log)
    codebook_encode_list_tail_kernel<<<tiles < grid_tail ? tiles : grid_tail, 512, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_listed_f32 launch");
}

static int encode_launch(const qv2x_encode_desc* d, const int8_t* in, const float* in_f32, const float* const* level_weights, uint8_t* codes, void* stream,
                         int form) {
    using namespace qv2x;
    if (!d || !in || !level_weights || !codes) return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 4) return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: bad shape");
    const int segs = d->segs ? d->segs : 1;                            // (0: a descriptor from before the field existed)
    if (segs != 1 && segs != 2 && segs != 4) return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: seg_num must be 1, 2 or 4 (got %d)", d->segs);
    if (d->kc < 32 || d->kc > 256 || d->kc % 32 || (segs > 1 && d->kc % 64) || segs * d->kc > 32 * ENC_MAX_TILES)
        return fail(QV2X_EINVAL, "qv2x_codebook_encode_f32: dict_size must be a multiple of 32 up to 256 (of 64 with seg_num > 1; seg_num * dict_size <= %d): "
                                 "got dict_size %d, seg_num %d", 32 * ENC_MAX_TILES, d->kc, segs);
    if ((uintptr_t)in & 15) return fail(QV2X_EALIGN, "qv2x_codebook_encode_f32: in must be 16-byte aligned");
    EncArgs a;
    a.in = in; a.in_f32 = in_f32; a.codes = codes; a.n = d->n; a.h = d->h; a.w = d->w; a.levels = d->levels; a.kc = d->kc;
    a.segs = segs; a.ke = segs * d->kc;
    const bool ext = segs > 1 || a.ke > 128;                           // (the workgroup form's general variant: encode_rows<ER, true>)
    a.ax = 128 - d->in_zx; a.dx = d->in_delta; a.M = d->n * d->h * d->w; a.m_lo = 0; a.m_hi = a.M;
    for (int l = 0; l < 4; ++l) {
        a.lvl[l] = l < d->levels ? level_weights[l] : nullptr;
        if (l < d->levels && (!a.lvl[l] || ((uintptr_t)a.lvl[l] & 15))) return fail(QV2X_EALIGN, "qv2x_codebook_encode_f32: level %d weights null or unaligned", l);
    }
    int er_env = 0;
#ifdef QV2X_DEV_KNOBS                                                  // dev builds only: 32 | 64 | 96 (mixed) rows per workgroup, anything else is refused
    static const int er_knob = [] { const char* e = getenv("QV2X_ENC_ROWS"); return e ? atoi(e) : 0; }();
    if (er_knob != 0 && er_knob != 32 && er_knob != 64 && er_knob != 96) return fail(QV2X_EINVAL, "QV2X_ENC_ROWS=%d: 32, 64 or 96", er_knob);
    er_env = er_knob;
#endif
    static std::atomic<int> cus_of[64];                                // CU count per device, asked once (0 = not asked yet); relaxed atomics: every
                                                                       // thread that finds 0 asks and stores the same value (reentrant per stream)
    int cus = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        cus = cus_of[dev].load(std::memory_order_relaxed);
        if (cus == 0) {
            cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
            cus_of[dev].store(cus, std::memory_order_relaxed);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (form == 1) return encode_wave_launch(a, st);
    // The wave form (codebook_encode_wave.hip: a wave per 32 cells, one wave per SIMD) runs whole rounds of the chip's 4 x CUs wave slots
    // at 0.88 of the f32 peak; a remainder of up to two workgroups per CU is the workgroup form's (32 cells per 8-wave workgroup: the same
    // cells in a third of a wave's time), a larger one is one more round of waves.  One V2X-Real frame = 1100 waves: one round of 1024 +
    // 76 workgroups, 410 us against 449 for the workgroup form alone; two frames 704 against 801, five 1643 against 1787
    // (profiles/r04_encode_by_frames.log).  Launches of less than one round keep the workgroup form's 64- / 32-row mix below.
    const int slots = 4 * cus, waves = (a.M + 31) / 32;
    const int rounds = waves / slots, rest = waves - rounds * slots;
    int tail_max = 2 * cus;                                            // most workgroups a remainder may have
#ifdef QV2X_DEV_KNOBS
    static const int tail_knob = [] { const char* e = getenv("QV2X_ENC_TAIL"); return e ? atoi(e) : -1; }();   // 0: every remainder as waves
    if (tail_knob >= 0) tail_max = tail_knob;
#endif
    if (er_env == 0 && rounds >= 1) {
        if (rest == 0 || rest > tail_max) return encode_wave_launch(a, st);
        EncArgs main = a, tail = a;
        main.m_hi = rounds * slots * 32;
        tail.m_lo = main.m_hi;
        if (int rc = encode_wave_launch(main, st)) return rc;
        if (ext) codebook_encode_kernel<32, true><<<rest, 512, 0, st>>>(tail);
        else codebook_encode_kernel<32><<<rest, 512, 0, st>>>(tail);
        return hip_check(hipGetLastError(), "qv2x_codebook_encode_f32 launch");
    }
    // Below one round of waves (a.M < 128 cells per CU): the mixed form when it has at least one full round of 64-row workgroups, else
    // 32-row workgroups.  (A whole launch of 64-row workgroups is a dev knob only since the split rule above: QV2X_ENC_ROWS=64.)
    const int n64 = (a.M / 64) / cus * cus;
    const int er = er_env ? er_env : (n64 > 0 ? 96 : 32);
    if (er == 64) {
        if (ext) codebook_encode_kernel<64, true><<<(a.M + 63) / 64, 512, 0, st>>>(a);
        else codebook_encode_kernel<64><<<(a.M + 63) / 64, 512, 0, st>>>(a);
    } else if (er == 96 && n64 > 0) {
        if (ext) codebook_encode_mixed_kernel<true><<<n64 + (a.M - n64 * 64 + 31) / 32, 512, 0, st>>>(a, n64);
        else codebook_encode_mixed_kernel<false><<<n64 + (a.M - n64 * 64 + 31) / 32, 512, 0, st>>>(a, n64);
    } else if (ext) {
        codebook_encode_kernel<32, true><<<(a.M + 31) / 32, 512, 0, st>>>(a);
    } else {
        codebook_encode_kernel<32><<<(a.M + 31) / 32, 512, 0, st>>>(a);
    }
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_f32 launch");
}

#ifdef QV2X_ENC_FINE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_encode_fine(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(qv2x::g_enc_fine), (size_t)n * 40 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef QV2X_ENC_TRACE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_encode_blocks(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(qv2x::g_enc_blk), (size_t)n * 4 * sizeof(unsigned long long));
}
#endif
