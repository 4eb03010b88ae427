// a6, stage 1 of the two-stage EXACT encode (quantv2x_amd/encode_two_stage.py has the derivation; include/qv2x.h the contract):
// UMGMQuantizer.encode (opencood/models/sub_modules/codebook.py:330-337 -> :231-239 -> :106-131) with its affine heads multiplied out,
//     dist_l[k] - |q_l|^2 = s_l[k] = G_l[k] . x_0 + g_l[k] + sum_{j<l} T_lj[code_j][k],        x_0 = delta (code - zx),
// evaluated WITHOUT ROUNDING: G on a fixed-point grid h (24 bits as three balanced int8 limbs), the contraction with the cell's 256 stored
// bytes on v_mfma_i32_32x32x32_i8 (exact i32 sums), the limbs, the bias and the table rows combined in fp64 on integers below 2^53.  Per
// level the wave keeps the best and the second-best packed score 128 S + k of every cell; a cell whose gap is not larger than the bound
//     tau_l / h = t0 + t1 N0 + t2 N0^2 + sum |code - zx|,        N0 = delta sqrt(sum (code - zx)^2)
// at ANY level is appended to the list stage 2 (codebook_encode_wave_kernel in list mode) recomputes in the reference's op order; the
// others keep these indices, which the bound proves to be the strict minimum of the fp32 chain too.
//
// One wave = 64 cells (two 32-cell B tiles: every A fragment feeds two MFMAs) through all levels; the products are transposed as in
// codebook_encode_wave.hip -- A = 32 scores x 32 input channels of one limb, streamed from L2 in fragment order by buffer loads with a
// scalar running offset; B = the cells' bytes exactly as the padded i8 map stores them (code - 128: the offset is folded into the bias),
// 64 registers for the whole kernel -- so a lane holds ONE cell (lane & 31) and 16 of a tile's 32 scores: the running minimum stays in the
// lane, one exchange between the half-waves closes a level.  No LDS, no barrier.  36 fragments x 8 steps x 2 = 576 MFMAs per 64 cells
// (0.59 MOP per cell against the chain's 43.8 MFLOP per agent-frame / 35 200 cells = 1.25 MFLOP per cell in fp32).
#include "common.h"

namespace qv2x {
namespace {

struct CandArgs {
    const int8_t* in; const int8_t* gpack; const int2* bias; const short* tables;
    uint8_t* codes; unsigned* list; unsigned* counters;
    float tau[3][3];
    int n, h, w, hw, M, levels, kc, zx, gbytes, tbytes, tshift;
    float delta;
};

constexpr int LIMBS = 3;
constexpr int FR = LIMBS * 8;                                        // A fragments (1 KiB each) of one tile of 32 scores
template <int V> struct IC { static constexpr int value = V; };
constexpr int CT = 2;                                                // 32-cell B tiles per wave: every A fragment feeds CT MFMAs
constexpr int NWV = 8;                                               // waves per workgroup: two per SIMD (a lone wave issues its i8 MFMAs at half the pipe's rate)
constexpr int NTH = 64 * NWV;

// LDS of a workgroup (four waves = 512 cells): the tile's A fragments, double-buffered, and the level's residual tables.  The four waves
// walk the same tiles in the same order, so ONE copy of the 24 KiB a tile's fragments take serves all of them (first version: every wave
// streamed them through the CU's L1 by itself, and gathered its cells' table rows from L2 -- 168 KB per tile and CU against the 64 B / clock
// the vector-memory path returns: the same-weights ablation ran 1.7x faster, profiles/r06_cand_ablations.log).
struct CandLds {
    v4i afrag[2][FR][64];                                            // 48 KiB
    int2 bias[3 * 128];                                              // every level's biases as (bias >> 16, bias & 0xffff): read per group of scores from LDS
    short tab1[128 * 128];                                           // 32 KiB: T_10 (level 1's table), int16 at 2^tshift grid units, 16-byte pieces XOR-swizzled by row
    short tab2[2 * 128 * 128];                                       // 64 KiB: T_20 | T_21 (level 2's), [kc][kc] each, back to back; each level's tables are copied in
                                                                     // behind the MFMAs of the level BEFORE it
};

__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void encode_candidates_kernel(const CandArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    CandLds& s = *(CandLds*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, hf = lane >> 5;
    const int m0 = ((int)blockIdx.x * NWV + wave) * (32 * CT);         // (a wave past the last cell still walks the tiles: the barriers are the workgroup's)

    // ---- the wave's 128 cells: B fragments straight from the padded map, and the two sums the bound needs ---------------------------------
    v4i xb[CT][8];
    int m[CT];
    float n0[CT], n1[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        m[ct] = m0 + 32 * ct + j;
        const int mc = m[ct] < a.M ? m[ct] : a.M - 1;
        const int img = mc / a.hw, rem = mc - img * a.hw, y = rem / a.w, x = rem - y * a.w;
        const int8_t* px = a.in + ((size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + x + 1) * 256 + 16 * hf;
#pragma unroll
        for (int k = 0; k < 8; ++k) xb[ct][k] = *(const v4i*)(px + 32 * k);
        unsigned s1 = 0, s2 = 0, sad = 0;
        const unsigned zx4 = (unsigned)a.zx * 0x01010101u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned u = (unsigned)xb[ct][k][d] ^ 0x80808080u;   // stored byte = code - 128
                s1 = __builtin_amdgcn_udot4(u, 0x01010101u, s1, false);
                s2 = __builtin_amdgcn_udot4(u, u, s2, false);
                sad = __builtin_amdgcn_sad_u8(u, zx4, sad);
            }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); sad += __shfl_xor(sad, 32);
        const int n2 = (int)s2 - 2 * a.zx * (int)s1 + 256 * a.zx * a.zx;    // sum (code - zx)^2 <= 256 * 255^2 < 2^24: exact as a float
        n0[ct] = a.delta * sqrtf((float)n2);
        n1[ct] = (float)sad;
    }

    // everything else by buffer loads: a scalar offset per (level, tile, group) and one 32-bit lane offset -- no 64-bit per-lane address
    // arithmetic (a first version with plain pointers spilled 300 registers of addresses); a load past the end of a resource returns zeros
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gpack, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.tables, 0, a.tbytes, 0x00020000);
    // a tile's 24 KiB of fragments: six 16-byte pieces per thread, global -> registers -> LDS one tile AHEAD of its use
    constexpr int NSTG = FR * 64 / NTH;                              // 16-byte pieces per thread
    v4i stg[NSTG];
    auto fetch = [&](int tt) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NSTG; ++e) stg[e] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(grs, (tid + NTH * e) * 16, tt * (FR * 1024), 0);
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NSTG; ++e) (&s.afrag[buf][0][0])[tid + NTH * e] = stg[e];
    };
    fetch(0);
    commit(0);
    for (int i = tid; i < a.levels * a.kc; i += NTH) s.bias[i] = a.bias[i];

    int r0[CT] = {}, r1[CT] = {};                                    // the cells' table rows: code_0, code_1
    const int ppr = a.kc >> 3, swz = (ppr & (ppr - 1)) == 0 ? ppr - 1 : 3;   // 16-byte pieces per table row; the swizzle mask (kc 96: rows of 192 bytes spread by themselves)
    bool flag[CT] = {};
    const int ntile = a.kc >> 5;
    int tt = 0;                                                      // tile counter over the levels
    auto level = [&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        // the running two smallest P = floor(S / 65536) of every cell and the index of the smallest: all in i32 at the full VALU rate (a first
        // version packed 128 S + k into fp64 and took ~110 cycles per candidate -- v_cvt_f64_i32 / v_min_f64 / v_max_f64 run at a quarter
        // of the rate: two thirds of the kernel's time).  S = 65536 (a2 + bh) + [256 a1 + a0 + bl + tables]: the bracket stays below 2^31
        // (|a1|, |a0| <= 2^22, bl < 2^16, two table entries below 2^15 << tshift <= 2^28 each: encode_two_stage.py), so
        // P = a2 + bh + (bracket >> 16) EXACTLY.  A cell is accepted when P_second - P_best >= ceil(tau / 65536) + 1, which implies
        // S_second - S_best > tau (floor loses less than one unit of 65536 on either side); equal P are a gap of 0 and go to stage 2.
        int best[CT], second[CT], bidx[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { best[ct] = second[ct] = 1 << 30; bidx[ct] = 0; }
        v4i qb[2][2];                                                // a group's biases, requested one group ahead of their use
        auto issue = [&](int T, int g, int slot) __attribute__((always_inline)) {
            const v4i* p = (const v4i*)&s.bias[l * a.kc + 32 * T + 8 * g + 4 * hf];
            qb[slot][0] = p[0];
            qb[slot][1] = p[1];
        };
#pragma unroll 1
        for (int T = 0; T < ntile; ++T, ++tt) {
            __syncthreads();                                            // this tile's fragments are in afrag[tt & 1]; the other buffer and (T == 0) the
            fetch(tt + 1);                                              // previous level's tables are no longer read by anyone
            const int buf = tt & 1;
            v16i acc[LIMBS][CT];
            v4i ar[3];
            ar[0] = s.afrag[buf][0][lane];
            ar[1] = s.afrag[buf][1][lane];
            // the NEXT level's tables, one chunk of 2 048 16-byte pieces per tile: requested here, written to LDS behind this tile's MFMAs.  A row's
            // pieces are XOR-swizzled by the row (`swz`): the cells of a wave read DIFFERENT rows at the SAME column, and rows of 256 bytes would
            // put them all on the same banks (a first LDS version: 32-way conflicts on every table read, 13k cycles per tile)
            constexpr int NTQ = 2048 / NTH;
            v4i tq[NTQ];
            const int npieces = (l + 1) * a.kc * (a.kc >> 3);           // int16 [l + 1 tables][kc][kc] = (l + 1) kc rows of kc / 8 pieces
            const bool copying = l + 1 < a.levels && T * 2048 < npieces;
            if (copying) {
#pragma unroll
                for (int e = 0; e < NTQ; ++e)
                    tq[e] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(trs, (T * 2048 + tid + NTH * e) * 16, ((l + 1) * l / 2) * a.kc * a.kc * 2, 0);
            }
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                if (f + 2 < FR) ar[(f + 2) % 3] = s.afrag[buf][f + 2][lane];
                const v4i A = ar[f % 3];
#ifndef QV2X_CAND_ABL_NOMFMA                                          // (dev ablation, tools/build_variant.py: two VALU instructions in an MFMA's place)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if ((f & 7) == 0) {
                        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[f >> 3][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, xb[ct][0], z, 0, 0, 0);
                    } else acc[f >> 3][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, xb[ct][f & 7], acc[f >> 3][ct], 0, 0, 0);
                }
#else
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if ((f & 7) == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[f >> 3][ct][r] = A[r & 3] + xb[ct][0][r >> 2];
                    } else { acc[f >> 3][ct][f & 15] += A[0] ^ xb[ct][f & 7][1]; acc[f >> 3][ct][(f + 8) & 15] ^= A[1]; }
                }
#endif
                if (f == FR - 2) issue(T, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                      // (hipcc otherwise sinks every load to its first use)
            }
            if (copying) {
                short* dst = l == 0 ? s.tab1 : s.tab2;
#pragma unroll
                for (int e = 0; e < NTQ; ++e) {
                    const int piece = T * 2048 + tid + NTH * e;
                    if (piece < npieces) {
                        const int row = piece / ppr, col = piece - row * ppr;
                        ((v4i*)dst)[row * ppr + (col ^ (row & swz))] = tq[e];
                    }
                }
            }
            commit((tt + 1) & 1);
#ifdef QV2X_CAND_ABL_PHASEBAR                                            // (dev ablation: no wave multiplies while another requantizes)
            __syncthreads();
#endif
#ifdef QV2X_CAND_ABL_SLEEP                                               // (dev ablation: 1 024 idle cycles between the last MFMA and the first read of a result)
            __builtin_amdgcn_s_sleep(16);
#endif
            // ---- the tile's 32 scores of each cell: lane (j, hf) holds scores 32 T + 8 g + 4 hf + e, g = r >> 2, e = r & 3 -----------------
#ifdef QV2X_CAND_ABL_NOEPI                                               // (dev ablation: one candidate per accumulator instead of sixteen)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int P = acc[2][ct][0] + (((acc[1][ct][0] << 8) + acc[0][ct][0] + qb[0][0][1]) >> 16);
                second[ct] = min(second[ct], max(best[ct], P));
                bidx[ct] = P < best[ct] ? 32 * T + 4 * hf : bidx[ct];
                best[ct] = min(best[ct], P);
            }
#else
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) issue(T, g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const int sl = g & 1;
                const int kcol = 32 * T + 8 * g + 4 * hf;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    int t4[4] = {0, 0, 0, 0};                           // the cell's table entries of these four scores, in 2^tshift grid units
                    if (l >= 1) {
                        const int2 u = *(const int2*)&(l == 1 ? s.tab1 : s.tab2)[(r0[ct] * ppr + ((kcol >> 3) ^ (r0[ct] & swz))) * 8 + (kcol & 7)];
                        t4[0] += (short)(u.x & 0xffff); t4[1] += u.x >> 16; t4[2] += (short)(u.y & 0xffff); t4[3] += u.y >> 16;
                    }
                    if (l >= 2) {
                        const int2 u = *(const int2*)&s.tab2[((a.kc + r1[ct]) * ppr + ((kcol >> 3) ^ (r1[ct] & swz))) * 8 + (kcol & 7)];
                        t4[0] += (short)(u.x & 0xffff); t4[1] += u.x >> 16; t4[2] += (short)(u.y & 0xffff); t4[3] += u.y >> 16;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#ifdef QV2X_CAND_ABL_HALFREAD                                          // (dev ablation: every second accumulator register read twice)
                        const int r = 4 * g + (e & 2);
#else
                        const int r = 4 * g + e;
#endif
                        const int bh = qb[sl][e >> 1][2 * (e & 1)], bl = qb[sl][e >> 1][2 * (e & 1) + 1];
                        int lo = (acc[1][ct][r] << 8) + acc[0][ct][r] + bl;
                        if (l >= 1) lo += t4[e] << a.tshift;
                        const int P = acc[2][ct][r] + bh + (lo >> 16);
                        second[ct] = min(second[ct], max(best[ct], P));
                        bidx[ct] = P < best[ct] ? kcol + e : bidx[ct];
                        best[ct] = min(best[ct], P);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        // ---- close the level: the two half-waves' (best, second, index), the gap against the bound --------------------------------------------
        unsigned cnt = 0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int ob = __shfl_xor(best[ct], 32), os = __shfl_xor(second[ct], 32), oi = __shfl_xor(bidx[ct], 32);
            const int nb = min(best[ct], ob), ns = min(max(best[ct], ob), min(second[ct], os));
            const int c = (ob < best[ct] || (ob == best[ct] && oi < bidx[ct])) ? oi : bidx[ct];
            if (l == 0) r0[ct] = c;
            if (l == 1) r1[ct] = c;
            const float t = ((a.tau[l][0] + a.tau[l][1] * n0[ct]) + (a.tau[l][2] * n0[ct]) * n0[ct]) + n1[ct];
            const bool weak = !((float)(ns - nb) >= ceilf(t * (1.0f / 65536.0f)) + 1.0f);     // (not accepted; ns - nb < 2^31: exact as a float up to 2^24, monotone beyond)
            cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(weak && hf == 0 && m[ct] < a.M && !flag[ct]));
            flag[ct] = flag[ct] || weak;
            if (hf == 0 && m[ct] < a.M) a.codes[(size_t)l * a.M + m[ct]] = (uint8_t)c;
        }
        if (lane == 0 && cnt) atomicAdd(a.counters + 1 + l, cnt);       // statistics: cells FIRST flagged at level l
    };
#ifdef QV2X_CAND_ABL_STOP                                            // dev ablation (tools/build_variant.py): 1 = stop after the setup, 2 = after level 0
    if (QV2X_CAND_ABL_STOP == 1) { if (hf == 0 && m[0] < a.M) a.codes[m[0]] = (uint8_t)(n0[0] + n1[1] + xb[0][3][1] + xb[1][7][2]); return; }
#endif
    level(IC<0>{});
#ifdef QV2X_CAND_ABL_STOP
    if (QV2X_CAND_ABL_STOP == 2) return;
#endif
    if (a.levels > 1) level(IC<1>{});
    if (a.levels > 2) level(IC<2>{});
    // ---- the cells stage 2 recomputes, in no particular order ------------------------------------------------------------------------------
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const bool mine = flag[ct] && hf == 0 && m[ct] < a.M;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine);
        if (mask) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(a.counters, (unsigned)__builtin_popcountll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (mine) a.list[base + (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (unsigned)m[ct];
        }
    }
}

// (a kernel, not hipMemsetAsync: every clear of this library is a kernel node under stream capture)
__global__ void zero_counters_kernel(unsigned* c) { if (threadIdx.x < 4) c[threadIdx.x] = 0; }

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_codebook_encode_candidates_i8(const qv2x_encode_desc* d, const int8_t* in, const int8_t* g_limbs, const int32_t* bias_split,
                                                  const int16_t* tables, int table_shift, const float* tau, uint8_t* codes, uint32_t* list,
                                                  uint32_t* counters, void* stream) {
    using namespace qv2x;
    if (!d || !in || !g_limbs || !bias_split || !tau || !codes || !list || !counters) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 3) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: 1..3 levels");
    if (d->segs > 1) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: seg_num 1 only (the exact entry takes seg_num 1 | 2 | 4)");
    if (d->kc < 32 || d->kc > 128 || d->kc % 32) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: dict_size must be 32, 64, 96 or 128 (got %d)", d->kc);
    if (d->levels > 1 && !tables) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: the residual levels need their tables");
    if (d->in_zx < 0 || d->in_zx > 255) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: in_zx outside 0..255");
    if (table_shift < 0 || table_shift > 13) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: table_shift 0..13 (two int16 entries << shift are added in i32)");
    if (((uintptr_t)in & 15) || ((uintptr_t)g_limbs & 15) || ((uintptr_t)tables & 15) || ((uintptr_t)bias_split & 7))
        return fail(QV2X_EALIGN, "qv2x_codebook_encode_candidates_i8: 16-byte aligned maps, limbs and tables");
    CandArgs a;
    a.in = in; a.gpack = g_limbs; a.bias = (const int2*)bias_split; a.tables = tables; a.codes = codes; a.list = list; a.counters = counters;
    for (int l = 0; l < 3; ++l)
        for (int i = 0; i < 3; ++i) a.tau[l][i] = l < d->levels ? tau[l * 3 + i] : 0.0f;
    a.n = d->n; a.h = d->h; a.w = d->w; a.hw = d->h * d->w; a.M = d->n * a.hw; a.levels = d->levels; a.kc = d->kc; a.zx = d->in_zx;
    a.gbytes = d->levels * (d->kc / 32) * LIMBS * 8 * 1024;
    a.tbytes = (d->levels > 1 ? d->levels * (d->levels - 1) / 2 : 1) * d->kc * d->kc * 2;
    a.tshift = table_shift;
    a.delta = d->in_delta;
    if (int rc = hip_check(hipFuncSetAttribute((const void*)encode_candidates_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CandLds)),
                           "qv2x_codebook_encode_candidates_i8: LDS size")) return rc;
    zero_counters_kernel<<<1, 64, 0, (hipStream_t)stream>>>(counters);
    encode_candidates_kernel<<<(a.M + 32 * CT * NWV - 1) / (32 * CT * NWV), NTH, sizeof(CandLds), (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_candidates_i8 launch");
}
