// a6, the form that whole rounds of the chip's wave slots take (encode_launch, codebook_encode.hip): ONE WAVE owns 32 BEV cells through
// every level; nothing is shared between waves.
//
// The workgroup form (codebook_encode.hip) computes out[cells][channels] = in . W^T with the eight waves of a workgroup splitting the
// output channels, so every GEMM of the chain ends with a tile store, a workgroup barrier and a re-read, and the |q|^2 / argmin phases
// cross waves through LDS: its four GEMM phases per level run at the matrix pipe's rate, the phases between them are the missing 20 %
// (profiles/r04_enc_fine_b32.log), and weaving them into the GEMMs does not pay with two waves per SIMD.
//
// Here the product is transposed: D[channel][cell] = W[channel][k] . act[k][cell].  The weights are the MFMA's A operand, streamed from
// L2 in fragment order (one 16-byte buffer load per lane feeds four MFMAs); the activations of the wave's 32 cells are the B operand and stay in
// 128 registers per matrix; a pair of 32-channel output tiles (two 16-register accumulators, two dependent chains of 128 MFMAs) goes through the wave's
// own 33 KB of LDS to come back in B-operand order -- no workgroup barrier, no other wave involved.  In the C layout a lane holds ONE cell
// (lane & 31) and 16 channels / codes per tile, so the argmin over the codes is a running minimum inside the lane plus one exchange
// between the two half-waves, and |q|^2 is two 64-long chains per lane read back from the wave's LDS rows.  seg_num (m) > 1
// (codebook.py:115-131): the codebook comes EXTENDED -- [m * kc][256], segment s in dims [s d, (s + 1) d), zeros elsewhere -- so the
// distance GEMM is the same stream over m * kc rows, a segment's argmin closes after its kc rows, and the code planes are [levels * m].  One wave per SIMD (256
// VGPRs + ~220 AGPRs, no scratch), four single-wave workgroups per CU (LDS), 32-cell scheduling granularity.  Measured: DESIGN.md §3.
//
// Bit-exactness: every dot product is the same ascending-k fp32 fma chain with acc0 = bias (v_mfma_f32_32x32x2_f32 adds k = 2t from
// lanes 0-31, then k = 2t + 1 from lanes 32-63; a . b commutes), |q|^2 and the distance use the workgroup form's op order, ties go to
// the lower code: identical codes (tests/test_hip_encode_wave.py against the oracle and against the workgroup form).
#include "codebook_encode.h"

namespace qv2x {

namespace {

constexpr int D = ENC_D;
constexpr int RS = 260;                 // LDS floats per cell row: b128 reads of 16 lanes land on 16 distinct 16-byte slots, b64 writes of 32 lanes on all banks
constexpr int NPF = 4;                  // weight groups (4 k-steps x 64 lanes x 4 B) in flight ahead of the MFMAs

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Within every group of eight consecutive k the wave's LDS rows hold (k0, k2, k4, k6, k1, k3, k5, k7): half-wave h feeds MFMA step
// t with k = 2t + h, so its four steps of a group are one float4.

// C layout -> LDS: register r of tile T is channel 32 T + 8 (r >> 2) + 4 h + (r & 3) of cell j
__device__ __forceinline__ void tile_to_lds(float* __restrict__ row, int h, int T, const v16f& acc) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        float* g = row + (4 * T + rq) * 8 + 2 * h;
        *(float2*)g = make_float2(acc[4 * rq], acc[4 * rq + 2]);
        *(float2*)(g + 4) = make_float2(acc[4 * rq + 1], acc[4 * rq + 3]);
    }
}

// LDS -> B-operand registers: m[4 g + s] = channel 8 g + 2 s + h of cell j
__device__ __forceinline__ void lds_to_operand(const float* __restrict__ row, int h, float (&m)[128]) {
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        const v4f v = *(const v4f*)(row + g * 8 + 4 * h);
        m[4 * g] = v.x; m[4 * g + 1] = v.y; m[4 * g + 2] = v.z; m[4 * g + 3] = v.w;
    }
}

// the four float4 of a per-channel (or per-code) vector that cover this lane's 16 registers of tile T
__device__ __forceinline__ void tile_vec(const float* __restrict__ p, int h, int T, v4f (&o)[4]) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) o[rq] = *(const v4f*)(p + 32 * T + 8 * rq + 4 * h);
}

#ifdef QV2X_ENCW_FINE   // dev build (tools/encw_fine.py): s_memtime stamps of the first 4096 waves at the phase boundaries of every level
__device__ long long g_encw_fine[4096][32];
#define WFINE(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_encw_fine[blockIdx.x][8 * l + (k)] = __builtin_readcyclecounter(); } while (0)
#define WFINE2(k) do { if (fine2 && threadIdx.x == 0 && blockIdx.x < 4096) g_encw_fine[blockIdx.x][24 + (k)] = __builtin_readcyclecounter(); } while (0)
#define WFINE2_ARM(v) fine2 = (v)
#else
#define WFINE(k) do { } while (0)
#define WFINE2(k) do { } while (0)
#define WFINE2_ARM(v) do { } while (0)
#endif

}  // namespace

template <int V> struct IC { static constexpr int value = V; };

// SEGS = seg_num (m).  SEGS > 1 (round 5): the distances walk only the codebook's DIAGONAL blocks -- a code of segment s is zero outside the
// 256 / m dims of its segment, and fma(x, 0, acc) = acc, so the sums are the dense walk's bit for bit with 1 / m of its MFMAs.  A tile pair
// then is one 32-code tile of each of TWO segments (tile 0 against operand registers [OFF0, ...), tile 1 against [OFF1, ...)), 32 / m groups
// long; the blob's codebook stream is packed in that order (engine.py:_pack_wave_seg, include/qv2x.h).
// LIST (round 6, stage 2 of the two-stage exact encode -- encode_two_stage.py): the wave's cells are `a.list[32 tile + j]`, tile = blockIdx.x,
// blockIdx.x + gridDim.x, ... below ceil(*a.list_count / 32) -- a persistent launch of fixed size whose work is a DEVICE-side count (the cells
// the candidate stage could not decide), so the whole encode stays capturable in a HIP graph.  The arithmetic is untouched.
template <bool F32IN, int SEGS, bool LIST = false>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(1, 1))) void codebook_encode_wave_kernel(const EncArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[32 * RS];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    float* const row = smem + j * RS;
    ListPlan plan{};
    if (LIST) plan = list_plan(a);
    // LIST: rounds of gridDim.x tiles, every second round walked BACKWARDS -- the tiles are numbered list by list (dearest first: a tile of
    // list 0 runs the whole chain, of list 2 seven of its eleven products), so a wave that took a dear tile in one round takes a cheap one in
    // the next (forwards only: the first waves got list 0 + list 1, the last list 1 + list 2)
#pragma unroll 1
    for (int round = 0; LIST ? round * (int)gridDim.x < plan.full : round == 0; ++round) {
    const int tile = LIST ? round * (int)gridDim.x + ((round & 1) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) : (int)blockIdx.x;
    if (LIST && tile >= plan.full) continue;
    int mrow;                                                        // this lane's cell, and whether its codes are stored
    bool owned;
    int start = 0;                                                   // LIST: the cells' codes below this level are PROVEN (the candidate stage accepted them):
    if (LIST) {                                                      // their quantization head, |q|^2 and distances are skipped, the stored code is used
        int cls, i0;
        list_tile(plan, tile, cls, i0);
        owned = i0 + j < plan.n[cls];
        mrow = (int)a.list[(size_t)cls * a.M + (owned ? i0 + j : plan.n[cls] - 1)];
        start = cls;
    } else {
        mrow = a.m_lo + tile * 32 + j;
        owned = mrow < a.m_hi;
    }

    float xq[128], z[128];              // B operands: x, then q, then the next x | z (read by qhead and by lhead)
    {   // ---- the wave's 32 rows of the BEV map, dequantized, straight into B-operand order -------------------------------
        int m = mrow;
        m = m < a.M ? m : a.M - 1;
        const int img = m / (a.h * a.w), rem = m - img * (a.h * a.w);
        const int y = rem / a.w, x = rem - y * a.w;
        const size_t pixel = (size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1;
        if (F32IN) {
            const v4f* sf = (const v4f*)(a.in_f32 + pixel * D);
#pragma unroll
            for (int q = 0; q < 64; ++q) {                                 // channels 4 q .. 4 q + 3: steps 2 q (k = 4 q + h) and 2 q + 1 (k = 4 q + 2 + h)
                const v4f v = sf[q];
                xq[2 * q] = h ? v.y : v.x;
                xq[2 * q + 1] = h ? v.w : v.z;
            }
        } else {
            const v4i* src = (const v4i*)(a.in + pixel * D);
            const float fax = (float)(a.ax - 128);
#pragma unroll
            for (int c = 0; c < 16; ++c) {                                 // 16 channels per load: groups 2 c and 2 c + 1
                const v4i raw = src[c];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // the stored byte is code - 128: code = byte ^ 0x80 as an unsigned byte; (float)(xs + ax) * dx with xs + ax = code + (ax - 128), exact in fp32
                    const unsigned u = ((unsigned)raw[d] ^ 0x80808080u) >> (8 * h);
                    xq[8 * c + 2 * d] = ((float)(u & 0xffu) + fax) * a.dx;
                    xq[8 * c + 2 * d + 1] = ((float)((u >> 16) & 0xffu) + fax) * a.dx;
                }
            }
        }
    }

    for (int l = 0; l < a.levels; ++l) {
        const float* W = a.lvl[l];
        const float* stage_b = W + D * D;
        const float* qhead_b = stage_b + D + D * D;
        const float* lhead_b = qhead_b + D + D * D;
        const float* cb = lhead_b + D + (size_t)D * a.ke;              // [ke][256]: the extended codebook (ke = segs * kc rows)
        const float* c2 = cb + (size_t)a.ke * D;                        // [ke]
        // the wave section: stage | qhead | codebook | lhead, each [tile pair][32 groups][2 tiles][64 lanes][4 steps] -- ONE linear stream
        // (buffer loads: the section as a resource, a SCALAR running offset, the lane's constant 16-byte offset -- no per-lane 64-bit address
        // arithmetic between the MFMAs)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(W + level_floats_wg(a.ke)), 0, 0x7fffffff, 0x00020000);
        int wo = 0;                                                     // bytes into the section: one tile pair = 64 KiB
        const int loff = lane * 16;
        auto wload = [&](int grp) __attribute__((always_inline)) {
            return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, wo + grp * 1024, 0));
        };
        const bool last = l + 1 == a.levels;
        const int npair = (a.ke + 63) >> 6;                             // (an odd number of 32-code tiles: the last pair's second tile is zeros)

        // bias / |C|^2 of the NEXT tile pair: requested NPF groups before the current pair ends, i.e. BEFORE the ring loads that are still in
        // flight when the pair starts -- the in-order vmcnt wait for it leaves the whole ring in flight
        v4f bn[2][4];
        auto next_vec = [&](const float* p, int P) __attribute__((always_inline)) {
            tile_vec(p, h, 2 * P, bn[0]);
            tile_vec(p, h, 2 * P + 1, bn[1]);
        };
        auto next_c2 = [&](int P) __attribute__((always_inline)) {    // codes past the dictionary: |C|^2 = +inf, never the minimum
            tile_vec(c2, h, 2 * P, bn[0]);
            if (64 * P + 32 < a.ke) tile_vec(c2, h, 2 * P + 1, bn[1]);
            else
#pragma unroll
                for (int i = 0; i < 4; ++i) bn[1][i] = v4f{INFINITY, INFINITY, INFINITY, INFINITY};
        };
        // SEGS > 1: |C|^2 of code tile P of segments s0 (tile 0) and s1 (tile 1): rows s * kc + 32 P ... of the extended codebook
        auto next_c2s = [&](int s0, int s1, int P) __attribute__((always_inline)) {
            tile_vec(c2 + s0 * a.kc, h, P, bn[0]);
            tile_vec(c2 + s1 * a.kc, h, P, bn[1]);
        };
        next_vec(stage_b, 0);
        __builtin_amdgcn_sched_barrier(0);
        v4f ring[2 * NPF];
#pragma unroll
        for (int i = 0; i < 2 * NPF; ++i) ring[i] = wload(i);
        __builtin_amdgcn_sched_barrier(0);

        // two 32-channel tiles at once: acc0 = bn (BIAS) or 0, 2 x 128 MFMAs against the operand registers `m` -- TWO dependent chains, so
        // that the matrix pipe never waits for a result (one chain of v_mfma_f32_32x32x2_f32 on a lone wave: 0.875 of the rate,
        // tools/probes/mfma_f32_chain_probe.hip); the ring slots a group leaves are refilled at once; `hook` = the requests for the next pair
        v16f acc[2];
#ifdef QV2X_ENCW_FINE
        bool fine2 = false;
#endif
        auto tile2g = [&](const float (&m)[128], const bool biased, auto&& hook, auto ng_c, auto off0_c, auto off1_c) __attribute__((always_inline)) {
            constexpr int NG = decltype(ng_c)::value, OFF0 = decltype(off0_c)::value, OFF1 = decltype(off1_c)::value;
            WFINE2(0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = biased ? bn[t][r >> 2][r & 3] : 0.0f;
            WFINE2(1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const v4f A0 = ring[(2 * g) % (2 * NPF)], A1 = ring[(2 * g + 1) % (2 * NPF)];
                // (sched_barrier after every step: hipcc otherwise groups the four MFMAs of ONE accumulator -- a dependent chain again)
                acc[0] = mfma(A0.x, m[OFF0 + 4 * g], acc[0]);
                acc[1] = mfma(A1.x, m[OFF1 + 4 * g], acc[1]);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = mfma(A0.y, m[OFF0 + 4 * g + 1], acc[0]);
                acc[1] = mfma(A1.y, m[OFF1 + 4 * g + 1], acc[1]);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = mfma(A0.z, m[OFF0 + 4 * g + 2], acc[0]);
                acc[1] = mfma(A1.z, m[OFF1 + 4 * g + 2], acc[1]);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = mfma(A0.w, m[OFF0 + 4 * g + 3], acc[0]);
                acc[1] = mfma(A1.w, m[OFF1 + 4 * g + 3], acc[1]);
                if (g == NG - NPF - 1) hook();
                if (NG == 32 && g == 7) WFINE2(2);
                if (NG == 32 && g == 15) WFINE2(3);
                if (NG == 32 && g == 23) WFINE2(4);
#ifndef QV2X_ENCW_ABL_NOLOAD                                              // (dev ablation: the ring is never refilled)
                ring[(2 * g) % (2 * NPF)] = wload(2 * (g + NPF));
                ring[(2 * g + 1) % (2 * NPF)] = wload(2 * (g + NPF) + 1);
#endif
                __builtin_amdgcn_sched_barrier(0);                      // (hipcc otherwise sinks every load to its first use)
            }
#ifndef QV2X_ENCW_ABL_SAMEW                                               // (dev ablation: every pair streams the same 64 KiB -- L1 / L2 latency out of the picture)
            wo += NG * 2048;
#endif
            WFINE2(5);
        };
        auto tile2 = [&](const float (&m)[128], const bool biased, auto&& hook) __attribute__((always_inline)) {
            tile2g(m, biased, hook, IC<32>{}, IC<0>{}, IC<0>{});
        };

        WFINE(0);
        // ---- z = stage(x) -------------------------------------------------------------------------------------------------
#pragma unroll 1
        for (int P = 0; P < 4; ++P) {
            WFINE2_ARM(l == 1 && P == 1);                               // (dev stamps: one pair of the stage GEMM of level 1 in detail)
            tile2(xq, true, [&]() __attribute__((always_inline)) { next_vec(P < 3 ? stage_b : qhead_b, P < 3 ? P + 1 : 0); });
            tile_to_lds(row, h, 2 * P, acc[0]);
            tile_to_lds(row, h, 2 * P + 1, acc[1]);
            WFINE2(6);
            WFINE2_ARM(false);
        }
        WFINE(1);
        lds_to_operand(row, h, z);
        WFINE(2);
        unsigned bcs = 0;                                               // the segments' codes, one byte each
        if (LIST && l < start) {
            // a level whose code the candidate stage PROVED: no quantization head, no |q|^2, no distances -- the weight stream jumps to the
            // latent head (stage | qhead | codebook | lhead, 64 KiB per tile pair), the ring is primed again, the stored code is the code
            wo = (8 + npair) * 65536;
            next_vec(lhead_b, 0);
#pragma unroll
            for (int i = 0; i < 2 * NPF; ++i) ring[i] = wload(i);
            bcs = a.codes[(size_t)l * a.M + mrow];
        } else {
        // ---- q = qhead(z) -------------------------------------------------------------------------------------------------
#pragma unroll 1
        for (int P = 0; P < 4; ++P) {
            tile2(z, true, [&]() __attribute__((always_inline)) { if (P < 3) next_vec(qhead_b, P + 1); else if (SEGS == 1) next_c2(0); else next_c2s(0, 1, 0); });
            tile_to_lds(row, h, 2 * P, acc[0]);
            tile_to_lds(row, h, 2 * P + 1, acc[1]);
        }
        WFINE(3);
        lds_to_operand(row, h, xq);
        // |q|^2: four 64-wide ascending fma chains per cell; half-wave h runs chains 2 h and 2 h + 1 on the LDS row.  One segment (m = 1):
        // (p0 + p1) + (p2 + p3); two segments of 128 dims: p0 + p1 | p2 + p3; four of 64: the chains themselves (codebook.py:115-121:
        // x.reshape(n, m, d), (x ** 2).sum(2); oracle/qv2x_oracle.c:sumsq_seg)
        float x2s[4];
        {
            const float* qh = row + 128 * h;
            float p[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float s = 0.0f;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const v4f e = *(const v4f*)(qh + (8 * c + g) * 8), o = *(const v4f*)(qh + (8 * c + g) * 8 + 4);
                    s = fmaf(e.x, e.x, s); s = fmaf(o.x, o.x, s); s = fmaf(e.y, e.y, s); s = fmaf(o.y, o.y, s);
                    s = fmaf(e.z, e.z, s); s = fmaf(o.z, o.z, s); s = fmaf(e.w, e.w, s); s = fmaf(o.w, o.w, s);
                }
                p[c] = s;
            }
            const float mine = p[0] + p[1];
            const float other = __shfl_xor(mine, 32);
            if (a.segs == 1) {
                x2s[0] = x2s[1] = x2s[2] = x2s[3] = mine + other;      // (fp32 addition commutes: both half-waves hold (p0 + p1) + (p2 + p3))
            } else if (a.segs == 2) {
                x2s[0] = h ? other : mine; x2s[1] = h ? mine : other; x2s[2] = x2s[3] = 0.0f;
            } else {
                const float o0 = __shfl_xor(p[0], 32), o1 = __shfl_xor(p[1], 32);
                x2s[0] = h ? o0 : p[0]; x2s[1] = h ? o1 : p[1]; x2s[2] = h ? p[0] : o0; x2s[3] = h ? p[1] : o1;
            }
        }
        WFINE(4);
        // ---- distances to the codes, 64 per pair, and the running first-argmin inside the lane; a pair lies inside ONE segment
        //      (kc % 64 == 0 when segs > 1), a segment's argmin closes with its last pair ------------------------------------------------
        if constexpr (SEGS == 1) {
        float best = INFINITY;
        int bc = 0;
        const int ppseg = npair / a.segs;                               // pairs per segment
#pragma unroll 1
        for (int P = 0; P < npair; ++P) {
            const int seg = a.segs == 1 ? 0 : P / ppseg, cbase = seg * a.kc;
            const float x2 = seg == 0 ? x2s[0] : (seg == 1 ? x2s[1] : (seg == 2 ? x2s[2] : x2s[3]));
            v4f c2t[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) c2t[t][i] = bn[t][i];
            tile2(xq, false, [&]() __attribute__((always_inline)) { if (P + 1 < npair) next_c2(P + 1); else if (!last) next_vec(lhead_b, 0); });
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {                             // codes ascend with (t, r) inside a lane: a strict < keeps the first
                    const float s = x2 + c2t[t][r >> 2][r & 3];
                    const float d = s - 2.0f * acc[t][r];
                    const int code = 64 * P + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3) - cbase;
                    const bool lt = d < best;
                    best = lt ? d : best;
                    bc = lt ? code : bc;
                }
            if (a.segs == 1 ? P + 1 == npair : (P + 1) % ppseg == 0) {  // the segment's last pair: the two half-waves' minima, ties to the lower code
                const float od = __shfl_xor(best, 32);
                const int oc = __shfl_xor(bc, 32);
                if (od < best || (od == best && oc < bc)) bc = oc;
                if (h == 0 && owned) a.codes[((size_t)l * a.segs + seg) * a.M + mrow] = (uint8_t)bc;
                bcs |= (unsigned)bc << (8 * seg);
                best = INFINITY; bc = 0;
            }
        }
        } else {
            // ---- SEGS > 1: a pair = code tile P of segments (sa, sb), 32 / SEGS groups; each segment its own running first-argmin ----------
            const int ntile = a.kc >> 5;                                // 32-code tiles per segment
            auto seg_pairs = [&](auto sa_c, auto sb_c, const bool more_after) __attribute__((always_inline)) {
                constexpr int SA = decltype(sa_c)::value, SB = decltype(sb_c)::value, NG = 32 / SEGS;
                float best[2] = {INFINITY, INFINITY};
                int bc[2] = {0, 0};
#pragma unroll 1
                for (int P = 0; P < ntile; ++P) {
                    v4f c2t[2][4];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) c2t[t][i] = bn[t][i];
                    tile2g(xq, false, [&]() __attribute__((always_inline)) {
                        if (P + 1 < ntile) next_c2s(SA, SB, P + 1);
                        else if (more_after) next_c2s(SA + 2, SB + 2, 0);
                        else if (!last) next_vec(lhead_b, 0);
                    }, IC<NG>{}, IC<4 * NG * SA>{}, IC<4 * NG * SB>{});
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float x2 = x2s[t ? SB : SA];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {                     // codes ascend with r inside a lane: a strict < keeps the first
                            const float sv = x2 + c2t[t][r >> 2][r & 3];
                            const float d = sv - 2.0f * acc[t][r];
                            const int code = 32 * P + 8 * (r >> 2) + 4 * h + (r & 3);
                            const bool lt = d < best[t];
                            best[t] = lt ? d : best[t];
                            bc[t] = lt ? code : bc[t];
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {                           // the two half-waves' minima, ties to the lower code
                    const int seg = t ? SB : SA;
                    const float od = __shfl_xor(best[t], 32);
                    const int oc = __shfl_xor(bc[t], 32);
                    if (od < best[t] || (od == best[t] && oc < bc[t])) bc[t] = oc;
                    if (h == 0 && owned) a.codes[((size_t)l * SEGS + seg) * a.M + mrow] = (uint8_t)bc[t];
                    bcs |= (unsigned)bc[t] << (8 * seg);
                }
            };
            if constexpr (SEGS == 2) seg_pairs(IC<0>{}, IC<1>{}, false);
            else { seg_pairs(IC<0>{}, IC<1>{}, true); seg_pairs(IC<2>{}, IC<3>{}, false); }
        }
        }                                                               // (the level's code was computed, not taken over)
        WFINE(5);
        if (last) break;
        // ---- x <- lhead(z) - C[code] ------------------------------------------------------------------------------------------
#pragma unroll 1
        for (int P = 0; P < 4; ++P) {
            // channels [64 P, 64 P + 64) lie in segment P * segs / 4: its codeword is row seg * kc + code of the extended codebook
            const int seg = (P * a.segs) >> 2;
            const float* cw = cb + (size_t)(seg * a.kc + (int)((bcs >> (8 * seg)) & 0xffu)) * D;
            v4f cv[2][4];
            tile_vec(cw, h, 2 * P, cv[0]);
            tile_vec(cw, h, 2 * P + 1, cv[1]);
            __builtin_amdgcn_sched_barrier(0);
            tile2(z, true, [&]() __attribute__((always_inline)) { if (P < 3) next_vec(lhead_b, P + 1); });
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] - cv[t][r >> 2][r & 3];
                tile_to_lds(row, h, 2 * P + t, acc[t]);
            }
        }
        WFINE(6);
        lds_to_operand(row, h, xq);
        WFINE(7);
    }
    }                                                                   // (the persistent loop of the LIST form)
}

int encode_wave_launch(const EncArgs& a, hipStream_t st) {
    const unsigned grid = (unsigned)((a.m_hi - a.m_lo + 31) / 32);
    if (a.segs == 2) {
        if (a.in_f32) codebook_encode_wave_kernel<true, 2><<<grid, 64, 0, st>>>(a);
        else codebook_encode_wave_kernel<false, 2><<<grid, 64, 0, st>>>(a);
    } else if (a.segs == 4) {
        if (a.in_f32) codebook_encode_wave_kernel<true, 4><<<grid, 64, 0, st>>>(a);
        else codebook_encode_wave_kernel<false, 4><<<grid, 64, 0, st>>>(a);
    } else if (a.in_f32) codebook_encode_wave_kernel<true, 1><<<grid, 64, 0, st>>>(a);
    else codebook_encode_wave_kernel<false, 1><<<grid, 64, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_f32 (wave form) launch");
}


// stage 2 of the two-stage encode: the listed cells, `waves` persistent single-wave workgroups (four per CU fill the chip)
int encode_wave_list_launch(const EncArgs& a, int waves, hipStream_t st) {
    if (a.segs != 1 || a.in_f32) return fail(QV2X_EINVAL, "qv2x_codebook_encode_listed_f32: seg_num 1, i8 rows");
    codebook_encode_wave_kernel<false, 1, true><<<(unsigned)waves, 64, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_listed_f32 launch");
}

}  // namespace qv2x

#ifdef QV2X_ENCW_FINE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_encw_fine(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(qv2x::g_encw_fine), (size_t)n * 32 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
