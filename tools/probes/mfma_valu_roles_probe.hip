// Do a pure-MFMA wave and a pure-VALU wave on the SAME SIMD run at full speed beside each other (gfx950)?  The question behind every
// "one wave multiplies while its SIMD-mate requantizes" design.  8-wave workgroups, one per CU: wave w and w + 4 share a SIMD.
//   mode 0: waves 0-3 run an i8 MFMA stream (5 independent accumulators), waves 4-7 idle       -> ticks per MFMA alone
//   mode 1: waves 4-7 run a VALU stream (8 independent chains of the epilogue's mix), 0-3 idle  -> ticks per VALU instruction alone
//   mode 2: both at once                                                                         -> what each costs beside the other
//   mode 3: all eight waves run MFMA; mode 4: all eight run VALU (the same-role pairs, for reference)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/roles tools/probes/mfma_valu_roles_probe.hip && /tmp/roles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

//   mode 5: as 2 with the roles swapped (the OLDER waves 0-3 run VALU); modes 6 / 7: as 2 with s_setprio 3 on the VALU / on the MFMA waves
template <int PRIO_VALU, int PRIO_MFMA>
__global__ __launch_bounds__(512, 1) void roles(int mode, int n, long long* out, int* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool do_mfma = (mode == 0 && wave < 4) || (mode == 2 && wave < 4) || mode == 3 || (mode == 5 && wave >= 4);
    const bool do_valu = (mode == 1 && wave >= 4) || (mode == 2 && wave >= 4) || mode == 4 || (mode == 5 && wave < 4);
    if (do_valu && PRIO_VALU) __builtin_amdgcn_s_setprio(PRIO_VALU);
    if (do_mfma && PRIO_MFMA) __builtin_amdgcn_s_setprio(PRIO_MFMA);
    long long t0 = 0, t1 = 0;
    int keep = 0;
    __syncthreads();
    if (do_mfma) {
        v16i acc[5];
        for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 7, 11, 13};
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 5; ++i) keep += acc[i][0] + acc[i][7];
    } else if (do_valu) {
        float f[8]; int q[8]; unsigned pk[8];
        for (int i = 0; i < 8; ++i) { f[i] = 1.0f + i + lane; q[i] = i * 977 + lane; pk[i] = 0; }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {                      // the requantizer's mix: mad_i24, cvt, mul, add, fma, fma, cvt_pk, cvt_pk
                q[i] = __mul24(q[i], 3) + it;
                float y = (float)q[i];
                y = y * 1.0001f; y = y + f[i];
                const float ta = __builtin_fmaf(y, 0.37f, 3.0001f), tb = __builtin_fmaf(y, 0.37f, 2.9999f);
                pk[i] = __builtin_amdgcn_cvt_pk_u8_f32(ta, i & 3, pk[i]);
                pk[(i + 1) & 7] = __builtin_amdgcn_cvt_pk_u8_f32(tb, i & 3, pk[(i + 1) & 7]);
                f[i] = ta * 1e-3f;
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) keep += (int)pk[i] + q[i] + (int)f[i];
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (keep == 123456789) sink[0] = keep;
}

int main() {
    long long* d; int* sink; hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 64);
    const int n = 2000;
    const char* names[8] = {"MFMA waves alone", "VALU waves alone", "MFMA waves 0-3 beside VALU waves 4-7", "all eight waves MFMA", "all eight waves VALU",
                            "VALU waves 0-3 (older) beside MFMA 4-7", "as 2, s_setprio 3 on the VALU waves", "as 2, s_setprio 3 on the MFMA waves"};
    for (int mode = 0; mode < 8; ++mode) {
        auto launch = [&]() {
            if (mode == 6) roles<3, 0><<<256, 512>>>(2, n, d, sink);
            else if (mode == 7) roles<0, 3><<<256, 512>>>(2, n, d, sink);
            else roles<0, 0><<<256, 512>>>(mode, n, d, sink);
        };
        launch();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0; int nm = 0, nv = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) { if (h[b * 8 + w] <= 0) continue; const bool is_m = mode == 3 || ((mode == 0 || mode == 2 || mode >= 6) && w < 4) || (mode == 5 && w >= 4); if (is_m) { m += h[b * 8 + w]; ++nm; } else { v += h[b * 8 + w]; ++nv; } }
        printf("%-40s %.3f ms;", names[mode], ms);
        if (nm) printf("  MFMA waves: %.1f ticks per MFMA (per wave)", m / nm / (10.0 * n));
        if (nv) printf("  VALU waves: %.2f ticks per VALU instruction (per wave; 64 per iteration)", v / nv / (64.0 * n));
        printf("\n");
    }
    return 0;
}
