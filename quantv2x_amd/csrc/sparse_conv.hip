// a13: the quantized SECOND encoder -- MeanVFE (opencood/models/sub_modules/mean_vfe.py:13-32), the twelve sparse 3-D convolutions of
// VoxelBackBone8x (sub_modules/sparse_backbone_3d.py:48-91) under QuantSpconvModule.forward (quant/quant_layer.py:460-490) and
// HeightCompression (sub_modules/height_compression.py:12-27).  The reference runs these on spconv (a third-party wheel that is not
// in the reference tree): the semantics are restated (oracle/spec_second.py), not bound.
//
// Data layout (sized for 288 GB: dense index volumes instead of hash tables):
//   * a level's active sites are ROWS: coords i32 [cap][4] = (agent, z, y, x), features i8 [cap + 1][C] holding (code - 128), C a
//     multiple of 32; row `cap` is the FILL row (zp - 128 in every channel = the real value 0).  The row count is a DEVICE int32:
//     nothing on this path waits for the host (the whole encoder is graph-capturable).
//   * per level one dense i32 volume [agents][D][H][W] maps a coordinate to its row (-1: no site).  At 0.1 m over 281.6 x 80 x 4 m
//     that is 369 MB per agent at level 1 -- 0.13 % of the HBM, against a hash probe per neighbour lookup.  It is cleared by
//     un-scattering the rows that were set, not by a memset.
//   * the rows of every level a SparseConv3d opens are in raster order of (agent, z, y, x) (mark -> scan -> assign over the volume).
//   * a convolution = rulebook + gather-GEMM.  The rulebook nbr i32 [K][cap_out] holds, per output site and window offset, the input
//     row or the fill row; sub-manifold layers that share an `indice_key` share it.  Absent neighbours read the fill row, so the
//     GEMM is the dense gemmlowp identity of the 2-D convolutions (T = sum xs*ws + aw*sum xs + corr) with no per-site bookkeeping.
//   * GEMM: the weights are the A operand of v_mfma_i32_32x32x32_i8 (staged once per workgroup in LDS, fragment order), the 32
//     gathered neighbour rows of a wave's tile are the B operand, so a lane owns one output site and 16 channels per 32-channel
//     tile in four runs of four: the epilogue (scale, BatchNorm1d affine, ReLU, requantize, 4-byte stores) is per lane.
//     Workgroups are persistent over 32-site tiles (8 waves share the up to 108 KB weight image).
#include "common.h"

namespace qv2x {
namespace {

__device__ __forceinline__ long long lin4(int b, int z, int y, int x, int D, int H, int W) {
    return (((long long)b * D + z) * H + y) * W + x;
}

__global__ void mean_vfe_kernel(const float* __restrict__ vox, const int32_t* __restrict__ nump, const int32_t* __restrict__ n_rows, int cap,
                                int T, float* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= min(*n_rows, cap)) return;
    const v4f* p = (const v4f*)(vox + (size_t)m * T * 4);
    v4f s = p[0];
    for (int t = 1; t < T; ++t) {                       // slot by slot, as the checker sums
        const v4f v = p[t];
        s[0] = s[0] + v[0]; s[1] = s[1] + v[1]; s[2] = s[2] + v[2]; s[3] = s[3] + v[3];
    }
    const float n = fmaxf((float)nump[m], 1.0f);
    v4f o = {s[0] / n, s[1] / n, s[2] / n, s[3] / n};
    *(v4f*)(out + (size_t)m * 4) = o;
}

// value >= 0: volume[coord(row)] = row;  value < 0: volume[coord(row)] = -1 (undo)
__global__ void index_scatter_kernel(const int32_t* __restrict__ coords, const int32_t* __restrict__ n_rows, int cap, int D, int H, int W,
                                     int32_t* __restrict__ vol, int set) {
    const int n = min(*n_rows, cap);
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
        const v4i c = *(const v4i*)(coords + (size_t)m * 4);
        vol[lin4(c[0], c[1], c[2], c[3], D, H, W)] = set ? m : -1;
    }
}

struct Geom {
    int k[3], s[3], p[3];
    int iD, iH, iW, oD, oH, oW;
};

// Active outputs of a strided sparse convolution, in RASTER order of (agent, z, y, x):
//   mark    every (input site, window offset) pair names at most one output position: its cell of the output volume is set to -2
//           (a plain idempotent store, no atomics);
//   count   marked cells per 1024-cell block of the volume;  scan: one workgroup turns the counts into block offsets and the total;
//   assign  every marked cell takes row = block offset + its rank inside the block, and writes its coordinates there.
// Raster rows are what makes the gather-GEMM cache-friendly (the 32 sites of a tile are neighbours along x and share most of their
// window), and they are deterministic -- the same order the CPU checker produces.
constexpr int SCAN_CELLS = 1024;                                   // cells per block of the count / assign passes (256 threads x 4)

__global__ void mark_sites_kernel(const int32_t* __restrict__ in_coords, const int32_t* __restrict__ n_in, int cap_in, const Geom g,
                                  int32_t* __restrict__ out_vol) {
    const int K = g.k[0] * g.k[1] * g.k[2];
    const long long total = (long long)min(*n_in, cap_in) * K;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(t / K), kk = (int)(t - (long long)m * K);
        const v4i c = *(const v4i*)(in_coords + (size_t)m * 4);
        const int kz = kk / (g.k[1] * g.k[2]), ky = (kk / g.k[2]) % g.k[1], kx = kk % g.k[2];
        const int nz = c[1] + g.p[0] - kz, ny = c[2] + g.p[1] - ky, nx = c[3] + g.p[2] - kx;
        if (nz < 0 || ny < 0 || nx < 0 || nz % g.s[0] || ny % g.s[1] || nx % g.s[2]) continue;
        const int oz = nz / g.s[0], oy = ny / g.s[1], ox = nx / g.s[2];
        if (oz >= g.oD || oy >= g.oH || ox >= g.oW) continue;
        out_vol[lin4(c[0], oz, oy, ox, g.oD, g.oH, g.oW)] = -2;
    }
}

__device__ __forceinline__ int block_sum_256(int v, int* sh) {       // sum over the 256 threads of a block; sh: 4 ints
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const int r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void count_marks_kernel(const int32_t* __restrict__ vol, long long cells, int32_t* __restrict__ counts) {
    __shared__ int sh[4];
    const long long base = (long long)blockIdx.x * SCAN_CELLS + threadIdx.x * 4;
    int c = 0;
    if (base + 3 < cells) {
        const v4i v = *(const v4i*)(vol + base);
        c = (v[0] == -2) + (v[1] == -2) + (v[2] == -2) + (v[3] == -2);
    } else {
        for (int e = 0; e < 4; ++e) c += (base + e < cells) && vol[base + e] == -2;
    }
    const int tot = block_sum_256(c, sh);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

// one workgroup: counts[nb] -> exclusive offsets in place, total -> *n_out (clamped to cap_out)
__global__ __launch_bounds__(1024) void scan_counts_kernel(int32_t* __restrict__ counts, int nb, int32_t* __restrict__ n_out, int cap_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? counts[i] : 0;
        int inc = v;                                                // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < nb) counts[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = min(carry_s, cap_out);
}

__global__ __launch_bounds__(256) void assign_rows_kernel(int32_t* __restrict__ vol, long long cells, const int32_t* __restrict__ offsets,
                                                          int oD, int oH, int oW, int32_t* __restrict__ out_coords, int cap_out) {
    __shared__ int wsum[4];
    const long long base = (long long)blockIdx.x * SCAN_CELLS + threadIdx.x * 4;
    int mk[4], c = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mk[e] = (base + e < cells) && vol[base + e] == -2; c += mk[e]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int row = offsets[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) row += wsum[w];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!mk[e]) continue;
        const long long cell = base + e;
        if (row < cap_out) {                                        // over capacity: dropped (the engine sizes cap_out so that it cannot happen)
            vol[cell] = row;
            const int x = (int)(cell % oW), y = (int)((cell / oW) % oH), z = (int)((cell / ((long long)oW * oH)) % oD);
            const int b = (int)(cell / ((long long)oW * oH * oD));
            v4i o = {b, z, y, x};
            *(v4i*)(out_coords + (size_t)row * 4) = o;
        } else {
            vol[cell] = -1;
        }
        ++row;
    }
}

__global__ void rulebook_kernel(const int32_t* __restrict__ out_coords, const int32_t* __restrict__ n_out, int cap_out, const Geom g,
                                const int32_t* __restrict__ in_vol, int fill_row, int32_t* __restrict__ nbr) {
    const int kk = blockIdx.y;
    const int kz = kk / (g.k[1] * g.k[2]), ky = (kk / g.k[2]) % g.k[1], kx = kk % g.k[2];
    const int n = min(*n_out, cap_out);
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
        const v4i c = *(const v4i*)(out_coords + (size_t)m * 4);
        const int z = c[1] * g.s[0] - g.p[0] + kz, y = c[2] * g.s[1] - g.p[1] + ky, x = c[3] * g.s[2] - g.p[2] + kx;
        int row = fill_row;
        if (z >= 0 && z < g.iD && y >= 0 && y < g.iH && x >= 0 && x < g.iW) {
            const int r = in_vol[lin4(c[0], z, y, x, g.iD, g.iH, g.iW)];
            if (r >= 0) row = r;
        }
        nbr[(size_t)kk * cap_out + m] = row;
    }
}

// Layer 0: fp32 voxel means (4 channels) in, 16 channels out.  One lane per site; acc = acc + x * w in window / channel order over the
// ACTIVE neighbours only (-ffp-contract=off: the multiply and the add round separately, as in the checker).
__global__ __launch_bounds__(256) void sp_conv_f32in_kernel(const float* __restrict__ feat, const int32_t* __restrict__ nbr,
                                                            const int32_t* __restrict__ n_out, int cap_out, int fill_row, int K,
                                                            const float* __restrict__ w, const float* __restrict__ bn_g,
                                                            const float* __restrict__ bn_h, float delta, float zp, int8_t* __restrict__ out) {
    __shared__ float wl[27 * 4 * 16];
    __shared__ float gl[16], hl[16];
    for (int i = threadIdx.x; i < K * 64; i += 256) wl[i] = w[i];
    if (threadIdx.x < 16) { gl[threadIdx.x] = bn_g[threadIdx.x]; hl[threadIdx.x] = bn_h[threadIdx.x]; }
    __syncthreads();
    const int n = min(*n_out, cap_out);
    const int m0 = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const bool valid = m0 < n;
    const int m = valid ? m0 : n - 1;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const int nb = nbr[(size_t)k * cap_out + m];
        if (nb == fill_row) continue;
        const v4f x = *(const v4f*)(feat + (size_t)nb * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = acc[co] + x[c] * wl[(k * 4 + c) * 16 + co];
    }
    const float rd = 1.0f / delta;
    int pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = 4 * q + e;
            float v = acc[co] * gl[co];
            v = v + hl[co];
            y[e] = fmaxf(v, 0.0f);
        }
        pk[q] = q_pack4_div(y[0], y[1], y[2], y[3], delta, rd, zp);
    }
    if (valid) {
        const int fillw = (int)(((unsigned)((int)zp - 128) & 255u) * 0x01010101u);
        v4i lo = {pk[0], pk[1], pk[2], pk[3]}, hi = {fillw, fillw, fillw, fillw};
        *(v4i*)(out + (size_t)m * 32) = lo;
        *(v4i*)(out + (size_t)m * 32 + 16) = hi;
    }
}

struct SpArgs {
    const int8_t* in; const int32_t* nbr; const int32_t* n_out; const int8_t* w; const float* scale; const int32_t* corr; const int32_t* aw;
    const float* bn_g; const float* bn_h; int8_t* out;
    int cap_out;
    float out_delta, out_zp;
};

// KOFF window offsets (27 or 3), KS = C_in / 32, NT = C_out / 32
template <int KOFF, int KS, int NT>
__global__ __launch_bounds__(512) void sp_conv_i8_kernel(const SpArgs a) {
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    constexpr int WBYTES = KOFF * KS * NT * 1024, CO = NT * 32, CI = KS * 32;
    const int n = min(*a.n_out, a.cap_out);
    if ((int)blockIdx.x * 256 >= n) return;
    int8_t* wl = smem;
    int* cst = (int*)(smem + WBYTES);                                // [aw | corr | scale | g | h][CO]
    for (int i = threadIdx.x; i < WBYTES / 16; i += 512) ((v4i*)wl)[i] = ((const v4i*)a.w)[i];
    for (int i = threadIdx.x; i < 5 * CO; i += 512) {
        const int which = i / CO, c = i - which * CO;
        cst[i] = which == 0 ? a.aw[c] : which == 1 ? a.corr[c] : which == 2 ? __float_as_int(a.scale[c])
               : which == 3 ? __float_as_int(a.bn_g[c]) : __float_as_int(a.bn_h[c]);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const float rd = 1.0f / a.out_delta;
    constexpr int G = 3, NG = KOFF / G;                              // offsets are gathered a group of three ahead of their MFMAs
    const int tiles = (n + 31) >> 5;
    for (int tile = blockIdx.x * 8 + wave; tile < tiles; tile += gridDim.x * 8) {
        const int m0 = tile * 32 + l31;
        const bool valid = m0 < n;
        const int m = valid ? m0 : n - 1;
        v16i acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0;
        int xs = 0;
        // a rolled loop over groups of three window offsets: the rows of group g + 1 are in flight during the MFMAs of group g, the
        // rulebook entries of group g + 2 behind them (fully unrolled, the compiler hoisted every weight read and spilled)
        int ni[G];
        v4i cur[G][KS], nxt[G][KS];
#pragma unroll
        for (int j = 0; j < G; ++j) ni[j] = a.nbr[(size_t)j * a.cap_out + m];
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) nxt[j][ks] = *(const v4i*)(a.in + (size_t)ni[j] * CI + ks * 32 + half * 16);
        if (NG > 1) {
#pragma unroll
            for (int j = 0; j < G; ++j) ni[j] = a.nbr[(size_t)(G + j) * a.cap_out + m];
        }
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) cur[j][ks] = nxt[j][ks];
            if (g + 1 < NG) {
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) nxt[j][ks] = *(const v4i*)(a.in + (size_t)ni[j] * CI + ks * 32 + half * 16);
                if (g + 2 < NG) {
#pragma unroll
                    for (int j = 0; j < G; ++j) ni[j] = a.nbr[(size_t)((g + 2) * G + j) * a.cap_out + m];
                }
            }
            const int8_t* wg = wl + (size_t)g * (G * KS * NT * 1024);
#pragma unroll
            for (int j = 0; j < G; ++j) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(cur[j][ks][q], 0x01010101, xs, false);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const v4i fa = *(const v4i*)(wg + ((j * KS + ks) * NT + t) * 1024 + lane * 16);
                        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, cur[j][ks], acc[t], 0, 0, 0);
                    }
                }
            }
        }
        const int tot = xs + __shfl_xor(xs, 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = t * 32 + 8 * q + 4 * half;
                const v4i xa = *(const v4i*)(cst + cl), xc = *(const v4i*)(cst + CO + cl);
                const v4f xsc = *(const v4f*)(cst + 2 * CO + cl), xg = *(const v4f*)(cst + 3 * CO + cl), xh = *(const v4f*)(cst + 4 * CO + cl);
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int T = acc[t][4 * q + e] + xa[e] * tot + xc[e];
                    float v = (float)T * xsc[e];
                    v = v * xg[e];
                    v = v + xh[e];
                    y[e] = fmaxf(v, 0.0f);
                }
                const int pk = q_pack4_div(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp);
                if (valid) *(int*)(a.out + (size_t)m * CO + cl) = pk;
                __builtin_amdgcn_sched_barrier(0);                  // keeps the constants of the next run of four from being hoisted (they spilled)
            }
        }
    }
}

template <int KOFF, int KS, int NT>
static int launch_sp(const SpArgs& a, hipStream_t st) {
    constexpr int LDS = KOFF * KS * NT * 1024 + 5 * NT * 32 * 4;
    // > 64 KB of dynamic LDS needs the attribute; set on every call (a host-side driver call: no process-global flag to go stale on a second device)
    if (int rc = hip_check(hipFuncSetAttribute((const void*)sp_conv_i8_kernel<KOFF, KS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), "qv2x_sp_conv_i8: LDS size")) return rc;
    const int per_cu = LDS > 80 * 1024 ? 1 : LDS > 52 * 1024 ? 2 : 3;            // 8-wave workgroups that fit a CU's 160 KB
    const int grid = max(1, min((a.cap_out + 255) / 256, 256 * per_cu));
    sp_conv_i8_kernel<KOFF, KS, NT><<<grid, 512, LDS, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_sp_conv_i8 launch");
}

// (a kernel, not hipMemsetAsync: inside a captured hipGraph the memset node did not stay ordered before the scatter on later replays)
__global__ void fill_bytes_kernel(v4i* __restrict__ p, size_t n16, int8_t* __restrict__ tail, int ntail, int word) {
    const v4i v = {word, word, word, word};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = (int8_t)word;
}

__global__ void to_bev_kernel(const int8_t* __restrict__ feat, const int32_t* __restrict__ coords, const int32_t* __restrict__ n_rows, int cap,
                              int C, int Cp, int D, int H, int W, int8_t* __restrict__ bev) {
    const int n = min(*n_rows, cap), c = threadIdx.x;
    if (c >= C) return;
    for (int m = blockIdx.x; m < n; m += gridDim.x) {
        const v4i p = *(const v4i*)(coords + (size_t)m * 4);
        bev[(((size_t)p[0] * (H + 2) + p[2] + 1) * (W + 2) + p[3] + 1) * (size_t)(C * D) + c * D + p[1]] = feat[(size_t)m * Cp + c];
    }
}

static Geom make_geom(const qv2x_spconv_desc* d) {
    Geom g;
    for (int i = 0; i < 3; ++i) { g.k[i] = d->k[i]; g.s[i] = d->s[i]; g.p[i] = d->p[i]; }
    g.iD = d->in_shape[0]; g.iH = d->in_shape[1]; g.iW = d->in_shape[2];
    g.oD = d->out_shape[0]; g.oH = d->out_shape[1]; g.oW = d->out_shape[2];
    return g;
}

static int check_geom(const qv2x_spconv_desc* d, const char* who) {
    if (!d) return fail(QV2X_EINVAL, "%s: null descriptor", who);
    for (int i = 0; i < 3; ++i) {
        if (d->k[i] < 1 || d->k[i] > 3 || d->s[i] < 1 || d->s[i] > 2 || d->p[i] < 0 || d->p[i] > 1) return fail(QV2X_EINVAL, "%s: window 1..3, stride 1..2, padding 0..1", who);
        if (d->in_shape[i] <= 0 || d->out_shape[i] <= 0) return fail(QV2X_EINVAL, "%s: empty volume", who);
        const int want = d->subm ? d->in_shape[i] : (d->in_shape[i] + 2 * d->p[i] - d->k[i]) / d->s[i] + 1;
        if (d->out_shape[i] != want) return fail(QV2X_EINVAL, "%s: out_shape[%d] = %d, the geometry gives %d", who, i, d->out_shape[i], want);
    }
    if (d->agents <= 0 || d->cap_in <= 0 || d->cap_out <= 0) return fail(QV2X_EINVAL, "%s: agents / capacities must be positive", who);
    return 0;
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_mean_vfe_f32(const float* voxel_features, const int32_t* voxel_num_points, const int32_t* n_voxels, int cap, int max_points,
                                 float* out, void* stream) {
    using namespace qv2x;
    if (!voxel_features || !voxel_num_points || !n_voxels || !out) return fail(QV2X_EINVAL, "qv2x_mean_vfe_f32: null pointer");
    if (cap <= 0 || max_points <= 0) return fail(QV2X_EINVAL, "qv2x_mean_vfe_f32: cap and max_points must be positive");
    if (((uintptr_t)voxel_features & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_mean_vfe_f32: 16-byte aligned pointers");
    mean_vfe_kernel<<<(cap + 255) / 256, 256, 0, (hipStream_t)stream>>>(voxel_features, voxel_num_points, n_voxels, cap, max_points, out);
    return hip_check(hipGetLastError(), "qv2x_mean_vfe_f32 launch");
}

extern "C" int qv2x_sp_index_scatter(const int32_t* coords, const int32_t* n_rows, int cap, int agents, int D, int H, int W, int32_t* volume,
                                     int set, void* stream) {
    using namespace qv2x;
    if (!coords || !n_rows || !volume) return fail(QV2X_EINVAL, "qv2x_sp_index_scatter: null pointer");
    if (cap <= 0 || agents <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(QV2X_EINVAL, "qv2x_sp_index_scatter: bad shape");
    if ((uintptr_t)coords & 15) return fail(QV2X_EALIGN, "qv2x_sp_index_scatter: 16-byte aligned coords");
    index_scatter_kernel<<<min((cap + 255) / 256, 2048), 256, 0, (hipStream_t)stream>>>(coords, n_rows, cap, D, H, W, volume, set);
    return hip_check(hipGetLastError(), "qv2x_sp_index_scatter launch");
}

extern "C" int64_t qv2x_sp_out_sites_workspace_bytes(const qv2x_spconv_desc* d) {
    using namespace qv2x;
    if (check_geom(d, "qv2x_sp_out_sites_workspace_bytes")) return -1;
    const long long cells = (long long)d->agents * d->out_shape[0] * d->out_shape[1] * d->out_shape[2];
    return ((cells + SCAN_CELLS - 1) / SCAN_CELLS) * 4 + 16;
}

extern "C" int qv2x_sp_out_sites(const qv2x_spconv_desc* d, const int32_t* in_coords, const int32_t* n_in, int32_t* out_volume,
                                 int32_t* out_coords, int32_t* n_out, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace qv2x;
    if (int e = check_geom(d, "qv2x_sp_out_sites")) return e;
    if (!in_coords || !n_in || !out_volume || !out_coords || !n_out || !workspace) return fail(QV2X_EINVAL, "qv2x_sp_out_sites: null pointer");
    if (d->subm) return fail(QV2X_EINVAL, "qv2x_sp_out_sites: a sub-manifold layer keeps its input's sites");
    if (((uintptr_t)in_coords & 15) || ((uintptr_t)out_coords & 15) || ((uintptr_t)out_volume & 15) || ((uintptr_t)workspace & 15))
        return fail(QV2X_EALIGN, "qv2x_sp_out_sites: 16-byte aligned pointers");
    if (workspace_bytes < qv2x_sp_out_sites_workspace_bytes(d)) return fail(QV2X_EINVAL, "qv2x_sp_out_sites: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const Geom g = make_geom(d);
    const long long cells = (long long)d->agents * g.oD * g.oH * g.oW;
    const int nb = (int)((cells + SCAN_CELLS - 1) / SCAN_CELLS);
    const long long threads = (long long)d->cap_in * g.k[0] * g.k[1] * g.k[2];
    mark_sites_kernel<<<(unsigned)min((threads + 255) / 256, 4096LL), 256, 0, st>>>(in_coords, n_in, d->cap_in, g, out_volume);
    count_marks_kernel<<<nb, 256, 0, st>>>(out_volume, cells, (int32_t*)workspace);
    scan_counts_kernel<<<1, 1024, 0, st>>>((int32_t*)workspace, nb, n_out, d->cap_out);
    assign_rows_kernel<<<nb, 256, 0, st>>>(out_volume, cells, (const int32_t*)workspace, g.oD, g.oH, g.oW, out_coords, d->cap_out);
    return hip_check(hipGetLastError(), "qv2x_sp_out_sites launch");
}

extern "C" int qv2x_sp_rulebook(const qv2x_spconv_desc* d, const int32_t* out_coords, const int32_t* n_out, const int32_t* in_volume,
                                int32_t* nbr, void* stream) {
    using namespace qv2x;
    if (int e = check_geom(d, "qv2x_sp_rulebook")) return e;
    if (!out_coords || !n_out || !in_volume || !nbr) return fail(QV2X_EINVAL, "qv2x_sp_rulebook: null pointer");
    if ((uintptr_t)out_coords & 15) return fail(QV2X_EALIGN, "qv2x_sp_rulebook: 16-byte aligned coords");
    const Geom g = make_geom(d);
    dim3 grid(min((d->cap_out + 255) / 256, 512), g.k[0] * g.k[1] * g.k[2]);
    rulebook_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(out_coords, n_out, d->cap_out, g, in_volume, d->cap_in, nbr);
    return hip_check(hipGetLastError(), "qv2x_sp_rulebook launch");
}

extern "C" int qv2x_sp_conv_f32in(const qv2x_spconv_desc* d, const float* feat, const int32_t* nbr, const int32_t* n_out, const float* w,
                                  const float* bn_g, const float* bn_h, int8_t* out, void* stream) {
    using namespace qv2x;
    if (int e = check_geom(d, "qv2x_sp_conv_f32in")) return e;
    if (!feat || !nbr || !n_out || !w || !bn_g || !bn_h || !out) return fail(QV2X_EINVAL, "qv2x_sp_conv_f32in: null pointer");
    if (d->cin != 4 || d->cout != 32) return fail(QV2X_EALIGN, "qv2x_sp_conv_f32in: 4 fp32 channels in, 16 (+16 pad) out");
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_sp_conv_f32in: out_delta must be positive");
    if (((uintptr_t)feat & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_sp_conv_f32in: 16-byte aligned pointers");
    const int K = d->k[0] * d->k[1] * d->k[2];
    sp_conv_f32in_kernel<<<(d->cap_out + 255) / 256, 256, 0, (hipStream_t)stream>>>(feat, nbr, n_out, d->cap_out, d->cap_in, K, w, bn_g, bn_h,
                                                                                  d->out_delta, d->out_zp, out);
    return hip_check(hipGetLastError(), "qv2x_sp_conv_f32in launch");
}

extern "C" int qv2x_sp_conv_i8(const qv2x_spconv_desc* d, const int8_t* in, const int32_t* nbr, const int32_t* n_out, const int8_t* w_frag,
                               const float* scale, const int32_t* corr, const int32_t* aw, const float* bn_g, const float* bn_h, int8_t* out,
                               void* stream) {
    using namespace qv2x;
    if (int e = check_geom(d, "qv2x_sp_conv_i8")) return e;
    if (!in || !nbr || !n_out || !w_frag || !scale || !corr || !aw || !bn_g || !bn_h || !out) return fail(QV2X_EINVAL, "qv2x_sp_conv_i8: null pointer");
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_sp_conv_i8: out_delta must be positive");
    if (((uintptr_t)in & 15) || ((uintptr_t)w_frag & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_sp_conv_i8: 16-byte aligned pointers");
    const int K = d->k[0] * d->k[1] * d->k[2];
    SpArgs a;
    a.in = in; a.nbr = nbr; a.n_out = n_out; a.w = w_frag; a.scale = scale; a.corr = corr; a.aw = aw; a.bn_g = bn_g; a.bn_h = bn_h; a.out = out;
    a.cap_out = d->cap_out; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    hipStream_t st = (hipStream_t)stream;
    const int ci = d->cin, co = d->cout;
    if (K == 27 && ci == 32 && co == 32) return launch_sp<27, 1, 1>(a, st);
    if (K == 27 && ci == 32 && co == 64) return launch_sp<27, 1, 2>(a, st);
    if (K == 27 && ci == 64 && co == 64) return launch_sp<27, 2, 2>(a, st);
    if (K == 3 && ci == 64 && co == 64) return launch_sp<3, 2, 2>(a, st);
    if (K == 3 && ci == 64 && co == 128) return launch_sp<3, 2, 4>(a, st);
    return fail(QV2X_EALIGN, "qv2x_sp_conv_i8: (window, cin, cout) = (%d, %d, %d) is not one of VoxelBackBone8x's layers", K, ci, co);
}

extern "C" int qv2x_sp_to_bev_i8(const int8_t* feat, const int32_t* coords, const int32_t* n_rows, int cap, int c, int c_padded, int agents, int D,
                                 int H, int W, int fill, int8_t* bev, void* stream) {
    using namespace qv2x;
    if (!feat || !coords || !n_rows || !bev) return fail(QV2X_EINVAL, "qv2x_sp_to_bev_i8: null pointer");
    if (cap <= 0 || c <= 0 || c > 1024 || c_padded < c || agents <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(QV2X_EINVAL, "qv2x_sp_to_bev_i8: bad shape");
    if (fill < -128 || fill > 127) return fail(QV2X_EINVAL, "qv2x_sp_to_bev_i8: fill is a (code - 128) byte");
    if ((uintptr_t)coords & 15) return fail(QV2X_EALIGN, "qv2x_sp_to_bev_i8: 16-byte aligned coords");
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)agents * (H + 2) * (W + 2) * c * D;
    if ((uintptr_t)bev & 15) return fail(QV2X_EALIGN, "qv2x_sp_to_bev_i8: 16-byte aligned map");
    const int word = (int)((unsigned)(fill & 255) * 0x01010101u);
    const size_t n16 = bytes / 16;
    fill_bytes_kernel<<<(unsigned)min((n16 + 255) / 256 + 1, (size_t)4096), 256, 0, st>>>((v4i*)bev, n16, bev + n16 * 16, (int)(bytes - n16 * 16), word);
    to_bev_kernel<<<min(cap, 8192), ((c + 63) / 64) * 64, 0, st>>>(feat, coords, n_rows, cap, c, c_padded, D, H, W, bev);
    return hip_check(hipGetLastError(), "qv2x_sp_to_bev_i8 launch");
}
