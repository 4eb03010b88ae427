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
// One wave = 128 cells at the batch, 64 for small launches (CT = four | two 32-cell B tiles: every A fragment feeds CT MFMAs) through all
// levels; the products are transposed as in codebook_encode_wave.hip -- A = 32 scores x 32 input channels of one limb, streamed from L2 in
// fragment order by buffer loads with a scalar running offset; B = the cells' bytes exactly as the padded i8 map stores them (code - 128:
// the offset is folded into the bias), 32 CT registers for the whole kernel -- so a lane holds ONE cell per tile (lane & 31) and 16 of a
// tile's 32 scores: the running minimum stays in the lane, one exchange between the half-waves closes a level.  No LDS and no barrier in the
// main loop (one at the end: the workgroup reserves its list ranges together).  36 fragments x 8 steps = 288 MFMAs per 32 cells (0.59 MOP per
// cell against the chain's 43.8 MFLOP per agent-frame / 35 200 cells = 1.25 MFLOP per cell in fp32).
#include "common.h"

namespace qv2x {
namespace {

struct CandArgs {
    const int8_t* in; const int8_t* gpack; const double* bias; const int* tables;
    uint8_t* codes; unsigned* list; unsigned* counters;
    float tau[3][3];
    int n, h, w, hw, M, levels, kc, zx, gbytes, tbytes;
    float delta;
};

constexpr int NPF = 8;                                               // A fragments (1 KiB each) in flight ahead of the MFMAs
constexpr int LIMBS = 3;
template <int V> struct IC { static constexpr int value = V; };

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __shfl_xor((int)b, m), hi = __shfl_xor((int)(b >> 32), m);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// CT = 32-cell B tiles per wave: every A fragment feeds CT MFMAs.  4 at the batch (the 288 KB of limbs a wave streams are shared by 128 cells);
// 2 where that launch would not reach every SIMD (one V2X-Real frame, 275 waves of 128 cells on 1024 SIMDs: 51.9 us; of 64 cells 33.2; of 32
// cells 40.9 -- the limbs are streamed per wave; four frames: 106.6 / 96.2 / 100.7).
template <int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void encode_candidates_kernel(const CandArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, hf = lane >> 5;
    const int m0 = ((int)blockIdx.x * 4 + wave) * (32 * CT);
    // (a wave past the last cell walks on with clamped cells and stores nothing: the workgroup meets at a barrier at the end)

    // ---- the wave's 64 cells: B fragments straight from the padded map, and the two sums the bound needs ---------------------------------
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
        for (int s = 0; s < 8; ++s) xb[ct][s] = *(const v4i*)(px + 32 * s);
        unsigned s1 = 0, s2 = 0, sad = 0;
        const unsigned zx4 = (unsigned)a.zx * 0x01010101u;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned u = (unsigned)xb[ct][s][d] ^ 0x80808080u;   // stored byte = code - 128
                s1 = __builtin_amdgcn_udot4(u, 0x01010101u, s1, false);
                s2 = __builtin_amdgcn_udot4(u, u, s2, false);
                sad = __builtin_amdgcn_sad_u8(u, zx4, sad);
            }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); sad += __shfl_xor(sad, 32);
        const int n2 = (int)s2 - 2 * a.zx * (int)s1 + 256 * a.zx * a.zx;    // sum (code - zx)^2 <= 256 * 255^2 < 2^24: exact as a float
        n0[ct] = a.delta * sqrtf((float)n2);
        n1[ct] = (float)sad;
    }

    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gpack, 0, a.gbytes, 0x00020000);   // past the end: zeros
    int wo = 0;
    const int loff = lane * 16;
    auto gload = [&](int f) __attribute__((always_inline)) { return (v4i)__builtin_amdgcn_raw_buffer_load_b128(grs, loff, wo + f * 1024, 0); };
    v4i ring[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) ring[i] = gload(i);

    // bias and table rows by buffer loads too: a scalar offset per (level, tile, group), ONE 32-bit lane offset per cell -- no 64-bit
    // per-lane address arithmetic (a first version with plain pointers spilled 300 registers of addresses)
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.levels * a.kc * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.tables, 0, a.tbytes, 0x00020000);
    int o0[CT] = {}, o1[CT] = {};                                    // byte offsets of the cells' table rows: (code_0 | code_1) * kc + 4 hf
    int first[CT];                                                   // the level a cell is first undecided at (3: never)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) first[ct] = 3;
    const int ntile = a.kc >> 5;
    auto level = [&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        double best[CT], second[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) best[ct] = second[ct] = 1.0e300;
        // a group's bias and table rows are requested one group AHEAD of their use (the first group's under the tile's last MFMAs): a wave is
        // alone on its SIMD, so a load issued where it is used costs its whole L2 round trip (first version: 878 us per 32 frames)
        v4i qb[2][2], qt0[2][CT], qt1[2][CT];
        auto issue = [&](int T, int g, int slot) __attribute__((always_inline)) {
            const int kofs = l * a.kc + 32 * T + 8 * g;                 // + 4 hf + e
            qb[slot][0] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(brs, 32 * hf, kofs * 8, 0);
            qb[slot][1] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(brs, 32 * hf + 16, kofs * 8, 0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (l >= 1) qt0[slot][ct] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(trs, o0[ct], ((l * (l - 1) / 2) * a.kc * a.kc + 32 * T + 8 * g) * 4, 0);
                if (l >= 2) qt1[slot][ct] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(trs, o1[ct], ((l * (l - 1) / 2 + 1) * a.kc * a.kc + 32 * T + 8 * g) * 4, 0);
            }
        };
#pragma unroll 1
        for (int T = 0; T < ntile; ++T) {
            v16i acc[LIMBS][CT];
#pragma unroll
            for (int lb = 0; lb < LIMBS; ++lb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[lb][ct][r] = 0;
#pragma unroll
            for (int f = 0; f < LIMBS * 8; ++f) {
                const v4i A = ring[f % NPF];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[f >> 3][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, xb[ct][f & 7], acc[f >> 3][ct], 0, 0, 0);
                ring[f % NPF] = gload(f + NPF);
                if (f == LIMBS * 8 - 6) issue(T, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                      // (hipcc otherwise sinks every load to its first use)
            }
            wo += LIMBS * 8 * 1024;
            // ---- the tile's 32 scores of each cell: lane (j, hf) holds scores 32 T + 8 g + 4 hf + e, g = r >> 2, e = r & 3 -----------------
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) issue(T, g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const int sl = g & 1;
                double b[4];                                            // 128 * bias + k: the packed form
                b[0] = __builtin_bit_cast(double, ((unsigned long long)(unsigned)qb[sl][0][1] << 32) | (unsigned)qb[sl][0][0]);
                b[1] = __builtin_bit_cast(double, ((unsigned long long)(unsigned)qb[sl][0][3] << 32) | (unsigned)qb[sl][0][2]);
                b[2] = __builtin_bit_cast(double, ((unsigned long long)(unsigned)qb[sl][1][1] << 32) | (unsigned)qb[sl][1][0]);
                b[3] = __builtin_bit_cast(double, ((unsigned long long)(unsigned)qb[sl][1][3] << 32) | (unsigned)qb[sl][1][2]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        // 128 S + k = 128 (65536 a2 + [256 a1 + a0 + tables] + bias) + k: the bracket in i32 (|a1| <= 2^22, |a0| <= 2^22, table
                        // entries below 2^28: encode_two_stage.py refuses larger ones), the rest on integers below 2^53 in fp64 -- all exact
                        int lo = (acc[1][ct][r] << 8) + acc[0][ct][r];
                        if (l >= 1) lo += qt0[sl][ct][e];
                        if (l >= 2) lo += qt1[sl][ct][e];
                        double c = __builtin_fma((double)lo, 128.0, b[e]);
                        c = __builtin_fma((double)acc[2][ct][r], 8388608.0, c);
                        second[ct] = fmin(second[ct], fmax(best[ct], c));
                        best[ct] = fmin(best[ct], c);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- close the level: the two half-waves' (best, second), the index out of the packed value, the gap against the bound ---------------
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const double ob = shfl_xor_f64(best[ct], 32), os = shfl_xor_f64(second[ct], 32);
            const double nb = fmin(best[ct], ob), ns = fmin(fmax(best[ct], ob), fmin(second[ct], os));
            const int c = (int)((long long)nb & 127);
            if (l == 0) o0[ct] = (c * a.kc + 4 * hf) * 4;
            if (l == 1) o1[ct] = (c * a.kc + 4 * hf) * 4;
            const float t = ((a.tau[l][0] + a.tau[l][1] * n0[ct]) + (a.tau[l][2] * n0[ct]) * n0[ct]) + n1[ct];
            const bool weak = !(ns - nb > 128.0 * (double)ceilf(t) + 127.0);      // (not accepted: the gap in S is <= ceil(t))
            if (weak && first[ct] == 3) first[ct] = l;
            if (hf == 0 && m[ct] < a.M) a.codes[(size_t)l * a.M + m[ct]] = (uint8_t)c;
        }
    };
    level(IC<0>{});
    if (a.levels > 1) level(IC<1>{});
    if (a.levels > 2) level(IC<2>{});
    // ---- the cells stage 2 recomputes, in THREE lists by the level they are first undecided at (list c at a.list + c M, its length in
    //      counters[1 + c], counters[0] = all of them): a cell of list c has PROVEN codes below level c, so stage 2 skips the quantization
    //      head and the distances of those levels for it (codebook_encode_wave.hip).  No particular order inside a list. ------------------
    // (ONE returning atomic per list and wave, the three issued together: a returning atomic is a whole L2 round trip for a lone wave, and a
    //  first version with one per list and cell tile -- twelve, each behind a branch -- cost the kernel 15 %)
    unsigned long long mask[3][CT];
    unsigned cnt[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            mask[c][ct] = __builtin_amdgcn_ballot_w64(first[ct] == c && hf == 0 && m[ct] < a.M);
            cnt[c] += (unsigned)__builtin_popcountll(mask[c][ct]);
        }
    // ... and one per list and WORKGROUP: the waves' counts meet in LDS, wave 0 reserves the workgroup's ranges
    __shared__ unsigned wcnt[4][3], wbase[3];
    if (lane == 0) { wcnt[wave][0] = cnt[0]; wcnt[wave][1] = cnt[1]; wcnt[wave][2] = cnt[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned tot = wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        wbase[threadIdx.x] = tot ? atomicAdd(a.counters + 1 + threadIdx.x, tot) : 0u;
        if (tot) atomicAdd(a.counters, tot);
    }
    __syncthreads();
    unsigned base[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        base[c] = wbase[c];
        for (int w2 = 0; w2 < wave; ++w2) base[c] += wcnt[w2][c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned at = __builtin_amdgcn_readfirstlane(base[c]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (first[ct] == c && hf == 0 && m[ct] < a.M)
                a.list[(size_t)c * a.M + at + (unsigned)__builtin_popcountll(mask[c][ct] & ((1ull << lane) - 1ull))] = (unsigned)m[ct];
            at += (unsigned)__builtin_popcountll(mask[c][ct]);
        }
    }
}

// (a kernel, not hipMemsetAsync: every clear of this library is a kernel node under stream capture)
__global__ void zero_counters_kernel(unsigned* c) { if (threadIdx.x < 4) c[threadIdx.x] = 0; }

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_codebook_encode_candidates_i8(const qv2x_encode_desc* d, const int8_t* in, const int8_t* g_limbs, const double* bias_packed,
                                                  const int32_t* tables, const float* tau, uint8_t* codes, uint32_t* list, uint32_t* counters,
                                                  void* stream) {
    using namespace qv2x;
    if (!d || !in || !g_limbs || !bias_packed || !tau || !codes || !list || !counters) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->levels < 1 || d->levels > 3) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: 1..3 levels");
    if (d->segs > 1) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: seg_num 1 only (the exact entry takes seg_num 1 | 2 | 4)");
    if (d->kc < 32 || d->kc > 128 || d->kc % 32) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: dict_size must be 32, 64, 96 or 128 (got %d)", d->kc);
    if (d->levels > 1 && !tables) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: the residual levels need their tables");
    if (d->in_zx < 0 || d->in_zx > 255) return fail(QV2X_EINVAL, "qv2x_codebook_encode_candidates_i8: in_zx outside 0..255");
    if (((uintptr_t)in & 15) || ((uintptr_t)g_limbs & 15) || ((uintptr_t)tables & 15) || ((uintptr_t)bias_packed & 7))
        return fail(QV2X_EALIGN, "qv2x_codebook_encode_candidates_i8: 16-byte aligned maps, limbs and tables");
    CandArgs a;
    a.in = in; a.gpack = g_limbs; a.bias = bias_packed; a.tables = tables; a.codes = codes; a.list = list; a.counters = counters;
    for (int l = 0; l < 3; ++l)
        for (int i = 0; i < 3; ++i) a.tau[l][i] = l < d->levels ? tau[l * 3 + i] : 0.0f;
    a.n = d->n; a.h = d->h; a.w = d->w; a.hw = d->h * d->w; a.M = d->n * a.hw; a.levels = d->levels; a.kc = d->kc; a.zx = d->in_zx;
    a.gbytes = d->levels * (d->kc / 32) * LIMBS * 8 * 1024;
    a.tbytes = (d->levels > 1 ? d->levels * (d->levels - 1) / 2 : 1) * d->kc * d->kc * 4;
    a.delta = d->in_delta;
    zero_counters_kernel<<<1, 64, 0, (hipStream_t)stream>>>(counters);
    int force_ct = 0;
#ifdef QV2X_DEV_KNOBS                                                  // dev builds only: 32-cell tiles per wave (2 | 4)
    static const int ct_env = getenv("QV2X_CAND_CT") ? atoi(getenv("QV2X_CAND_CT")) : 0;
    force_ct = ct_env == 2 || ct_env == 4 ? ct_env : 0;
#endif
    int dev = 0, cus = 256, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    const int ct = force_ct ? force_ct : (a.M >= 4 * cus * 128 * 2 ? 4 : 2);      // two rounds of waves of 128 cells, or the smaller tile
    if (ct == 4) encode_candidates_kernel<4><<<(a.M + 128 * 4 - 1) / (128 * 4), 256, 0, (hipStream_t)stream>>>(a);
    else encode_candidates_kernel<2><<<(a.M + 128 * 2 - 1) / (128 * 2), 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), "qv2x_codebook_encode_candidates_i8 launch");
}
