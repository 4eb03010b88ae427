// One instruction stream that alternates i8 MFMAs with the requantizer's VALU mix (the "woven epilogue"), on gfx950:
// how many cycles per MFMA, as a function of K = VALU instructions per MFMA, ONE chain vs FIVE independent accumulators,
// and ONE wave per SIMD (4-wave workgroups) vs TWO (8-wave workgroups)?  (mfma_valu_roles_probe: a pure-MFMA wave stalls its
// SIMD-mate's VALU completely -- the overlap has to happen inside one wave's stream.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/weave tools/probes/mfma_valu_weave_probe.hip && /tmp/weave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int K, int CH>      // K VALU per MFMA (multiples of 8: one "output" = 8 instructions), CH accumulator chains
__global__ __launch_bounds__(512, 1) void weave(int n, long long* out, int* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v16i acc[CH];
    for (int i = 0; i < CH; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 7, 11, 13};
    float f[4]; int q[4]; unsigned pk[4];
    for (int i = 0; i < 4; ++i) { f[i] = 1.0f + i + lane; q[i] = i * 977 + lane; pk[i] = 0; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            acc[i % CH] = __builtin_amdgcn_mfma_i32_32x32x32_i8(i & 1 ? b : a, i & 1 ? a : b, acc[i % CH], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K / 8; ++k) {                  // one output's requantization: mad_i24, cvt, mul, add, fma, fma, cvt_pk, cvt_pk
                const int j = (i + k) & 3;
                q[j] = __mul24(q[j], 3) + it;
                float y = (float)q[j];
                y = y * 1.0001f; y = y + f[j];
                const float ta = __builtin_fmaf(y, 0.37f, 3.0001f), tb = __builtin_fmaf(y, 0.37f, 2.9999f);
                pk[j] = __builtin_amdgcn_cvt_pk_u8_f32(ta, j, pk[j]);
                pk[(j + 1) & 3] = __builtin_amdgcn_cvt_pk_u8_f32(tb, j, pk[(j + 1) & 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int keep = 0;
    for (int i = 0; i < CH; ++i) keep += acc[i][0] + acc[i][7];
    for (int i = 0; i < 4; ++i) keep += (int)pk[i] + q[i];
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (keep == 123456789) sink[0] = keep;
}

template <int K, int CH> void run(long long* d, int* sink, int waves) {
    const int n = 1000;
    weave<K, CH><<<256, 64 * waves>>>(n, d, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    weave<K, CH><<<256, 64 * waves>>>(n, d, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) m += h[b * 8 + w];
    m /= 256.0 * waves * 10.0 * n;
    printf("K = %2d VALU per MFMA, %d chain(s), %d wave(s) per SIMD: %.3f ms, %.1f ticks per MFMA per wave = %.1f per MFMA per SIMD\n", K, CH, waves / 4, ms, m,
           m / (waves / 4));
}

int main() {
    long long* d; int* sink; hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 64);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0, 1>(d, sink, waves); run<0, 5>(d, sink, waves);
        run<8, 1>(d, sink, waves); run<8, 5>(d, sink, waves);
        run<16, 1>(d, sink, waves); run<16, 5>(d, sink, waves);
        run<24, 5>(d, sink, waves);
    }
    return 0;
}
