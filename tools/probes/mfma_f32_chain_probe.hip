// f32 MFMA rate by instruction shape, chains per wave and waves per SIMD (register operands only):
//   v_mfma_f32_32x32x2_f32  (2048 MACs, 16 passes): ONE dependent chain per wave -- what the exact codebook-encode kernel issues
//   v_mfma_f32_16x16x4_f32  (1024 MACs,  8 passes): FOUR independent chains per wave (a 32 x 32 wave tile as 2 x 2 tiles of 16 x 16)
// Both are bit-exact ascending-k fma chains (profiles/r01_mfma_probe.log).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/mfma_f32_chain_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SHAPE, int CHAINS>
__global__ void probe(float* out, int n, float seed) {
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    float s = 0.0f;
    if (SHAPE == 32) {
        v16f acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 8 / CHAINS; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        v4f acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = 0.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int CHAINS>
static void run(int waves_per_simd, float* out) {
    const int n = 20000, threads = 256 * waves_per_simd;              // one workgroup per CU, waves_per_simd waves on each of its 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<SHAPE, CHAINS><<<256, threads>>>(out, 10, 1.0f);
    hipEventRecord(e0);
    probe<SHAPE, CHAINS><<<256, threads>>>(out, n, 1.0f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double macs = 256.0 * 4 * waves_per_simd * n * 8 * 2048;    // every loop iteration: 8 x 2048 = 16 x 1024 MACs per wave
    printf("v_mfma_f32_%s  %d chain(s) per wave, %d wave(s) per SIMD: %7.3f ms  %6.1f TFLOP/s\n", SHAPE == 32 ? "32x32x2" : "16x16x4", CHAINS,
           waves_per_simd, ms, 2 * macs / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 4);
    for (int w : {1, 2, 4}) run<32, 1>(w, out);
    for (int w : {1, 2, 4}) run<32, 2>(w, out);
    for (int w : {1, 2, 4}) run<16, 1>(w, out);
    for (int w : {1, 2, 4}) run<16, 2>(w, out);
    for (int w : {1, 2, 4}) run<16, 4>(w, out);
    return 0;
}
