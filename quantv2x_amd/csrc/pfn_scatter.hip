// a1 + a2: pillar feature net (W8A8 semantics) fused with the scatter into the padded i8 BEV canvas.
// One wavefront per FOUR pillars, lane = output channel (64).  HBM-bound: reads M * (512 + 16 + 4) B, writes 64 B
// per pillar.  Every quantization step is monotone non-decreasing, so max over points commutes with it: the
// kernel takes the max of the pre-quantization value and quantizes once (bit-identical to the per-point form
// in oracle/qv2x_oracle.c:orc_pfn, verified by tests/test_hip_parity.py).
#include "common.h"

namespace qv2x {

// Round 4: FOUR pillars per wave, their headers and first point slots requested together (a pillar of a 60k-point sweep holds 2.2
// points on average): one wave per pillar was a chain of three dependent round trips (header -> points -> points again) per 64 bytes of
// output -- 268 us per batch of 32 sweeps, 0.27 of HBM.  Slots past the prefetched ones take the per-slot loop as before.
// Round 5: TWO prefetched slots and six workgroups per CU (209 -> 190-202 us; eight: the same time with 16 bytes of scratch; four slots 231, eight pillars per wave 216).
// What did NOT move it (profiles/r05_pfn_variants.log): the pillar mean by a table reciprocal + one exact correction step instead of three
// IEEE divisions, and the two quantizers as common.h's sandwich -- a third fewer VALU instructions, the same 197-199 us: the kernel waits
// for its scalar loads (34 s_load_dwordx4 per wave, lgkmcnt(0) each time -- scalar loads return out of order), not for the VALU.
constexpr int PFN_PB = 4, PFN_PQ = 2;

__global__ __launch_bounds__(256, 6) void pfn_scatter_kernel(const float4* __restrict__ vf, const int4* __restrict__ coords,
                                                          const int* __restrict__ npts, int M, int P,
                                                          const qv2x_pfn_params prm, int8_t* __restrict__ canvas,
                                                          int N, int ny, int nx) {
    const int lane = threadIdx.x & 63;
    int m0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PFN_PB;
    m0 = __builtin_amdgcn_readfirstlane(m0);
    if (m0 >= M) return;

    int4 c[PFN_PB];                           // (agent, z, y, x)
    int np[PFN_PB];
    float4 q[PFN_PB][PFN_PQ];
#pragma unroll
    for (int j = 0; j < PFN_PB; ++j) {
        const int m = m0 + j < M ? m0 + j : M - 1;
        c[j] = coords[m];
        np[j] = npts[m];
    }
#pragma unroll
    for (int j = 0; j < PFN_PB; ++j) {
        const int m = m0 + j < M ? m0 + j : M - 1;
#pragma unroll
        for (int p = 0; p < PFN_PQ; ++p) q[j][p] = vf[(size_t)m * P + (p < P ? p : P - 1)];
    }
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = prm.w[lane * 10 + k];
    const float b = prm.b[lane];

#pragma unroll
    for (int j = 0; j < PFN_PB; ++j) {
        if (m0 + j >= M) break;
        const float4* pts = vf + (size_t)(m0 + j) * P;
        // Ascending-slot sums.  The reference sums all P slots (pillar_vfe.py:118-119); the padded ones hold zeros -- the voxel
        // generator's contract, and the reference's own mean is wrong otherwise -- and x + 0.0f == x, so stopping at the point
        // count gives the same bits while reading ~2 slots per pillar instead of 32.
        const int real = np[j] < P ? np[j] : P;
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int p = 0; p < PFN_PQ; ++p)
            if (p < real) { sx += q[j][p].x; sy += q[j][p].y; sz += q[j][p].z; }
        for (int p = PFN_PQ; p < real; ++p) {
            const float4 t = pts[p];
            sx += t.x; sy += t.y; sz += t.z;
        }
        const float n = (float)np[j];
        const float mx = sx / n, my = sy / n, mz = sz / n;
        const float cx = (float)c[j].w * prm.vox[0] + prm.off[0];
        const float cy = (float)c[j].z * prm.vox[1] + prm.off[1];
        const float cz = (float)c[j].y * prm.vox[2] + prm.off[2];

        float ymax = -INFINITY;
        auto point = [&](const float4 t) __attribute__((always_inline)) {
            const float f[10] = {t.x, t.y, t.z, t.w, t.x - mx, t.y - my, t.z - mz, t.x - cx, t.y - cy, t.z - cz};
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 10; ++k) acc = fmaf(f[k], w[k], acc);
            ymax = fmaxf(ymax, acc + b);
        };
#pragma unroll
        for (int p = 0; p < PFN_PQ; ++p)
            if (p < real) point(q[j][p]);
        for (int p = PFN_PQ; p < real; ++p) point(pts[p]);
        if (real < P) ymax = fmaxf(ymax, b);       // zero-masked slots contribute the bias alone

        const float q1 = q_code(ymax, prm.d1, prm.z1);
        float y1 = (q1 - prm.z1) * prm.d1;
        y1 = fmaxf(y1, 0.0f);
        const int code = (int)q_code(y1, prm.d2, prm.z2);

        if (c[j].x < 0 || c[j].x >= N || c[j].z < 0 || c[j].z >= ny || (c[j].y + c[j].w) < 0 || (c[j].y + c[j].w) >= nx) continue;
        const size_t cell = ((size_t)c[j].x * (ny + 2) + (c[j].z + 1)) * (nx + 2) + (size_t)(c[j].y + c[j].w + 1);
        canvas[cell * 64 + lane] = (int8_t)(code - 128);
    }
}

// Round 6: SIXTEEN lanes per pillar, four output channels per lane.  The lane-per-channel form above spends ~80 of its ~100 vector
// instructions per pillar on values every lane of the wave shares (the three means and their divisions, the centre offsets, the ten
// features of a point, the cell address); here they are shared by four channels, a wave covers four pillars at once instead of one after
// the other, and the pillar data come by vector loads the wave can keep in flight (the scalar loads of the first form return out of
// order: lgkmcnt(0) after each).  Per (pillar, channel) the arithmetic is the same instruction sequence -- ascending-slot sums, ascending-k
// fma chain from 0.0f, + b, max, the two quantizers -- so the canvas is the same bit for bit (tests/test_hip_parity.py, test_hip_fullsize.py).
// PFN_R rounds of 16 pillars per workgroup, the next round's headers (coordinates, count, first PFN_NQ slots) requested one round ahead;
// the 40 weights of a lane come through LDS once.  217 -> 185 us per 32 sweeps (scatter + un-scatter; tools/bench_pfn.py).  What did NOT
// move it further, each bit-exact (profiles/r06_pfn_forms.log): 4 or 6 prefetched slots instead of 2 (185-187; 8: 343, registers); the second
// quantizer as the 256-entry table below and the first as common.h's sandwich (188 -> 185); every LANE fetching one pillar's header -- 64
// pillars per wave in one round trip -- and sixteen rounds taking them by ds_bpermute (232: 64 different lines per load instruction).  The
// rounds execute ~100 + 54 x (the LARGEST point count among the wave's four pillars) vector instructions: ~120 us of the 167 are the
// vector port, and what is left to take is the divergence between pillars of 1 and of 20 points (a flattened point list with a segmented
// max: not built -- the stage is 4 % of a step).
constexpr int PFN_R = 8;
#ifndef PFN_NQ
#define PFN_NQ 2
#endif

__global__ __launch_bounds__(256, 4) void pfn_scatter16_kernel(const float4* __restrict__ vf, const int4* __restrict__ coords,
                                                            const int* __restrict__ npts, int M, int P,
                                                            const qv2x_pfn_params prm, int8_t* __restrict__ canvas,
                                                            int N, int ny, int nx, int rounds) {
    __shared__ float wl[640 + 64];
    __shared__ uint8_t lut[256];
    for (int i = threadIdx.x; i < 640; i += 256) wl[i] = prm.w[i];
    if (threadIdx.x < 64) wl[640 + threadIdx.x] = prm.b[threadIdx.x];
    {   // the second quantizer sees one of 256 values -- dequantized code of the first, ReLU -- so it is a table: entry (code1 - 128) & 0xff
        // (the byte the first quantizer packs) holds (code2 - 128), computed by the very expressions of the lane-per-channel form
        const float q1 = (float)(threadIdx.x ^ 0x80);
        float y1 = (q1 - prm.z1) * prm.d1;
        y1 = fmaxf(y1, 0.0f);
        lut[threadIdx.x] = (uint8_t)((int)q_code(y1, prm.d2, prm.z2) - 128);
    }
    const float rd1 = 1.0f / prm.d1;
    __syncthreads();
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
    float w[4][10], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k = 0; k < 10; ++k) w[c][k] = wl[(sub * 4 + c) * 10 + k];
        b[c] = wl[640 + sub * 4 + c];
    }
    const int mbase = blockIdx.x * (16 * rounds) + slot;

    auto header = [&](int m, int4& c, int& np, float4 (&q)[PFN_NQ]) __attribute__((always_inline)) {
        const int mm = m < M ? m : M - 1;
        c = coords[mm];
        np = npts[mm];
#pragma unroll
        for (int p = 0; p < PFN_NQ; ++p) q[p] = vf[(size_t)mm * P + (p < P ? p : P - 1)];
    };
    int4 c, cn;
    int np, npn;
    float4 q[PFN_NQ], qn[PFN_NQ];
    header(mbase, c, np, q);
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int m = mbase + r * 16;
        if (r + 1 < rounds) header(m + 16, cn, npn, qn);
        if (m < M) {
            const float4* pts = vf + (size_t)m * P;
            const int real = np < P ? np : P;
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int p = 0; p < PFN_NQ; ++p)
                if (p < real) { sx += q[p].x; sy += q[p].y; sz += q[p].z; }
            for (int p = PFN_NQ; p < real; ++p) {
                const float4 t = pts[p];
                sx += t.x; sy += t.y; sz += t.z;
            }
            const float n = (float)np;
            const float mx = sx / n, my = sy / n, mz = sz / n;
            const float cx = (float)c.w * prm.vox[0] + prm.off[0];
            const float cy = (float)c.z * prm.vox[1] + prm.off[1];
            const float cz = (float)c.y * prm.vox[2] + prm.off[2];
            float ymax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            auto point = [&](const float4 t) __attribute__((always_inline)) {
                const float f[10] = {t.x, t.y, t.z, t.w, t.x - mx, t.y - my, t.z - mz, t.x - cx, t.y - cy, t.z - cz};
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 10; ++k) acc = fmaf(f[k], w[ch][k], acc);
                    ymax[ch] = fmaxf(ymax[ch], acc + b[ch]);
                }
            };
#pragma unroll
            for (int p = 0; p < PFN_NQ; ++p)
                if (p < real) point(q[p]);
            for (int p = PFN_NQ; p < real; ++p) point(pts[p]);
            if (real < P) {                               // zero-masked slots contribute the bias alone
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) ymax[ch] = fmaxf(ymax[ch], b[ch]);
            }
            // first quantizer: clamp(rint(y / d1) + z1, 0, 255) bit for bit (common.h: the division-exact sandwich), four codes in a dword
            const unsigned c1 = (unsigned)q_pack4_div(ymax[0], ymax[1], ymax[2], ymax[3], prm.d1, rd1, prm.z1);
            const unsigned word = (unsigned)lut[c1 & 0xff] | ((unsigned)lut[(c1 >> 8) & 0xff] << 8) | ((unsigned)lut[(c1 >> 16) & 0xff] << 16) |
                                  ((unsigned)lut[c1 >> 24] << 24);
            if (!(c.x < 0 || c.x >= N || c.z < 0 || c.z >= ny || (c.y + c.w) < 0 || (c.y + c.w) >= nx)) {
                const size_t cell = ((size_t)c.x * (ny + 2) + (c.z + 1)) * (nx + 2) + (size_t)(c.y + c.w + 1);
                *(unsigned*)(canvas + cell * 64 + sub * 4) = word;
            }
        }
        c = cn; np = npn;
#pragma unroll
        for (int p = 0; p < PFN_NQ; ++p) q[p] = qn[p];
    }
}

// The canvas stays resident and CLEAN between frames: instead of re-filling all of it before every scatter (9 MB per V2X-Real frame), the
// ~27k cells a frame's pillars wrote are set back to the code of 0.0 once the first convolution has read them.  One thread per (pillar,
// 16-byte piece); the same bounds test as the scatter.
__global__ __launch_bounds__(256) void pfn_unscatter_kernel(const int4* __restrict__ coords, int M, int value4, int8_t* __restrict__ canvas,
                                                            int N, int ny, int nx) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(id >> 2), piece = (int)(id & 3);
    if (m >= M) return;
    const int4 c = coords[m];
    if (c.x < 0 || c.x >= N || c.z < 0 || c.z >= ny || (c.y + c.w) < 0 || (c.y + c.w) >= nx) return;
    const size_t cell = ((size_t)c.x * (ny + 2) + (c.z + 1)) * (nx + 2) + (size_t)(c.y + c.w + 1);
    v4i v; v[0] = v[1] = v[2] = v[3] = value4;
    *(v4i*)(canvas + cell * 64 + piece * 16) = v;
}

}  // namespace qv2x

extern "C" int qv2x_pfn_unscatter_i8(const int32_t* voxel_coords, int M, int value, int8_t* canvas, int N, int ny, int nx, void* stream) {
    using namespace qv2x;
    if (M == 0) return QV2X_OK;
    if (!voxel_coords || !canvas) return fail(QV2X_EINVAL, "qv2x_pfn_unscatter_i8: null pointer");
    if (M < 0 || N <= 0 || ny <= 0 || nx <= 0 || value < -128 || value > 127) return fail(QV2X_EINVAL, "qv2x_pfn_unscatter_i8: bad sizes / value");
    if (((uintptr_t)voxel_coords & 15) || ((uintptr_t)canvas & 15)) return fail(QV2X_EALIGN, "qv2x_pfn_unscatter_i8: coords / canvas must be 16-byte aligned");
    const int b = value & 0xff, v4 = b | (b << 8) | (b << 16) | (b << 24);
    const long long threads = (long long)M * 4;
    pfn_unscatter_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>((const int4*)voxel_coords, M, v4, canvas, N, ny, nx);
    return hip_check(hipGetLastError(), "qv2x_pfn_unscatter_i8 launch");
}

extern "C" int qv2x_pfn_scatter_i8(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                                   int M, int max_points, const qv2x_pfn_params* params, int8_t* canvas, int N, int ny, int nx,
                                   void* stream) {
    using namespace qv2x;
    if (M == 0) return QV2X_OK;
    if (!voxel_features || !voxel_coords || !voxel_num_points || !params || !canvas)
        return fail(QV2X_EINVAL, "qv2x_pfn_scatter_i8: null pointer");
    if (M < 0 || max_points <= 0 || N <= 0 || ny <= 0 || nx <= 0) return fail(QV2X_EINVAL, "qv2x_pfn_scatter_i8: bad sizes");
    if (((uintptr_t)voxel_features & 15) || ((uintptr_t)voxel_coords & 15)) return fail(QV2X_EALIGN, "qv2x_pfn_scatter_i8: inputs must be 16-byte aligned");
    int form = 16;
#ifdef QV2X_DEV_KNOBS                                                  // dev builds only: 64 = the lane-per-channel form of rounds 1-5
    static const int form_env = getenv("QV2X_PFN_FORM") ? atoi(getenv("QV2X_PFN_FORM")) : 16;
    form = form_env;
#endif
    if (form == 64)
        pfn_scatter_kernel<<<(M + 4 * PFN_PB - 1) / (4 * PFN_PB), 256, 0, (hipStream_t)stream>>>((const float4*)voxel_features, (const int4*)voxel_coords,
                                                                         voxel_num_points, M, max_points, *params, canvas, N, ny, nx);
    else {
        // rounds of 16 pillars per workgroup: PFN_R at the batch (the weights' trip through LDS amortised), fewer for one sweep (27 k pillars:
        // eight rounds would leave 213 workgroups on 256 CUs -- 44 us against 14)
        static const int cus = [] {                                    // (one process drives one GPU model: asked once)
            int dev = 0, v = 0;
            return hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
        }();
        int rounds = M / (16 * 8 * cus);
        rounds = rounds < 1 ? 1 : (rounds > PFN_R ? PFN_R : rounds);
        pfn_scatter16_kernel<<<(M + 16 * rounds - 1) / (16 * rounds), 256, 0, (hipStream_t)stream>>>((const float4*)voxel_features, (const int4*)voxel_coords,
                                                                          voxel_num_points, M, max_points, *params, canvas, N, ny, nx, rounds);
    }
    return hip_check(hipGetLastError(), "qv2x_pfn_scatter_i8 launch");
}
