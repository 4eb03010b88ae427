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
    pfn_scatter_kernel<<<(M + 4 * PFN_PB - 1) / (4 * PFN_PB), 256, 0, (hipStream_t)stream>>>((const float4*)voxel_features, (const int4*)voxel_coords,
                                                                     voxel_num_points, M, max_points, *params, canvas, N, ny, nx);
    return hip_check(hipGetLastError(), "qv2x_pfn_scatter_i8 launch");
}
