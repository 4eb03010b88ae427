// How many requantizing-epilogue VALU instructions hide under one v_mfma_i32_32x32x32_i8 on gfx950?
// A loop of 5 MFMAs (independent accumulators, as one K half-step of the halo-patch conv) plus FILL x 5 epilogue-mix VALU instructions,
// either all MFMAs first and the VALU block after them (what hipcc emits for "K step; then some epilogue"), or one MFMA followed by
// FILL VALU instructions (sched_group_barrier).  1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/mfma_valu_overlap_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int FILL, bool INTER>
__global__ void probe(int* out, int n, int seed) {
    v16i acc[5];
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed ^ 5, seed ^ 6, seed ^ 7, seed ^ 9};
    // the epilogue's instruction mix on independent chains: T = mul24 + add3, cvt, mul, add, mul, rndne, sub, max3
    int ti[5]; float f[5], g[5], m = 0.0f;
    for (int i = 0; i < 5; ++i) { ti[i] = seed + i + threadIdx.x; f[i] = 1.0f + i; g[i] = 0.5f * i; }
    const float sc = 1.0001f, bs = 0.37f, rd = 0.731f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
            if (FILL >= 10) {   // one output's worth (~10 instructions) on chain i
                const int T = ti[i] + __mul24(ti[i], seed) + it;
                const float y = bs + (float)T * sc;
                const float t = y * rd;
                const float k = __builtin_rintf(t);
                m = __builtin_fmaxf(m, __builtin_fabsf(t - k));
                g[i] = __builtin_fminf(__builtin_fmaxf(k + 8388608.0f, 8388608.0f), 8388863.0f);
                ti[i] = T ^ __float_as_int(g[i]);
            }
            if (FILL >= 5 && FILL < 10) {
                const float y = bs + f[i] * sc;
                const float k = __builtin_rintf(y * rd);
                f[i] = __builtin_fmaxf(k - y, g[i]);
            }
            if (FILL >= 20) {
                const int T = ti[i] * 3 + __mul24(ti[i], seed ^ 77) + it;
                const float y = bs + (float)T * sc;
                const float t = y * rd;
                const float k = __builtin_rintf(t);
                m = __builtin_fmaxf(m, __builtin_fabsf(t - k));
                f[i] = __builtin_fminf(__builtin_fmaxf(k + 8388608.0f, 8388608.0f), 8388863.0f);
                ti[i] = T ^ __float_as_int(f[i]);
            }
        }
        if (INTER) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                if (FILL) __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);   // FILL VALU
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
            if (FILL) __builtin_amdgcn_sched_group_barrier(0x002, 5 * FILL, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 5; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; s += ti[i] + (int)f[i] + (int)g[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)m;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int)(t1 - t0);
}

template <int FILL, bool INTER>
void run(int* d, int waves) {
    const int n = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<FILL, INTER><<<256, 64 * waves>>>(d, n, 3);
    hipEventRecord(e0);
    probe<FILL, INTER><<<256, 64 * waves>>>(d, n, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    printf("fill %2d VALU per MFMA, %s, %d waves/SIMD: %.3f ms; wave 0: %.1f cycles per MFMA (%.1f per loop of 5)\n", FILL,
           INTER ? "interleaved 1:FILL" : "5 MFMAs then the VALU block", waves / 4, ms, (double)cyc / (n * 5), (double)cyc / n);
}

int main() {
    int* d; hipMalloc(&d, 1 << 22);
    for (int waves : {4, 8}) {
        run<0, false>(d, waves);
        run<5, false>(d, waves); run<5, true>(d, waves);
        run<10, false>(d, waves); run<10, true>(d, waves);
        run<20, false>(d, waves); run<20, true>(d, waves);
    }
    return 0;
}
